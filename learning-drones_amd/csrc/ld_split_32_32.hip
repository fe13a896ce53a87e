#define LD_TAG 32_32
#define LD_WIDTHS 32, 32
#include "ld_split.inc"
