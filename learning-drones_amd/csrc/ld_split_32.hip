#define LD_TAG 32
#define LD_WIDTHS 32
#include "ld_split.inc"
