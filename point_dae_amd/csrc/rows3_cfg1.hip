// rows3_cfg1.hip -- tile shape 1 of the exact-split row GEMM family: 64 x 128, 4 waves (rows3_cfg.inc)
#define R3_TI 1
#define R3_TJ 2
#define R3_WM 2
#define R3_WN 2
#define R3_KS 2
#define R3_NAME launch_rows3_cfg1
#include "rows3_cfg.inc"
