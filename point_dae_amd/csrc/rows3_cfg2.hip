// rows3_cfg2.hip -- tile shape 2 of the exact-split row GEMM family: 128 x 192, 8 waves (rows3_cfg.inc)
#define R3_TI 1
#define R3_TJ 3
#define R3_WM 4
#define R3_WN 2
#define R3_KS 2
#define R3_NAME launch_rows3_cfg2
#include "rows3_cfg.inc"
