#define LD_TAG 32
#define LD_WIDTHS 32
#define LD_SPLIT_KEEP_TU
#include "ld_split.inc"
