#define LD_TAG 10_10
#define LD_WIDTHS 10, 10
#include "ld_split.inc"
