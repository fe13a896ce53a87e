#define LD_TAG 16_16
#define LD_WIDTHS 16, 16
#include "ld_split.inc"
