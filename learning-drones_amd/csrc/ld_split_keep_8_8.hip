#define LD_TAG 8_8
#define LD_WIDTHS 8, 8
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
