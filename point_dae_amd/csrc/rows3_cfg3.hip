// rows3_cfg3.hip -- tile shape 3 of the exact-split row GEMM family: 128 x 64, 8 waves (rows3_cfg.inc)
#define R3_TI 1
#define R3_TJ 1
#define R3_WM 4
#define R3_WN 2
#define R3_KS 2
#define R3_NAME launch_rows3_cfg3
#include "rows3_cfg.inc"
