#define LD_TAG 10_10
#define LD_WIDTHS 10, 10
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
