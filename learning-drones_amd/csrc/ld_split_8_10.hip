#define LD_TAG 8_10
#define LD_WIDTHS 8, 10
#include "ld_split.inc"
