#define LD_TAG 64
#define LD_WIDTHS 64
#include "ld_split.inc"
