#define LD_TAG 24_24
#define LD_WIDTHS 24, 24
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
