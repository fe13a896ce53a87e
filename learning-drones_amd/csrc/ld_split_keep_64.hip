#define LD_TAG 64
#define LD_WIDTHS 64
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
