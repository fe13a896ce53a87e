#define LD_TAG 32_32
#define LD_WIDTHS 32, 32
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
