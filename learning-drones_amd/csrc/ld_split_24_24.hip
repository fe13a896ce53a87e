#define LD_TAG 24_24
#define LD_WIDTHS 24, 24
#include "ld_split.inc"
