#define LD_TAG 16_16
#define LD_WIDTHS 16, 16
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
