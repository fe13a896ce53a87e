#define LD_TAG 8_10
#define LD_WIDTHS 8, 10
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
