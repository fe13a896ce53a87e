#define LD_TAG 8_8
#define LD_WIDTHS 8, 8
#include "ld_split.inc"
