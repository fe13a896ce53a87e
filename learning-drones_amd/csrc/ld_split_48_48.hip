#define LD_TAG 48_48
#define LD_WIDTHS 48, 48
#include "ld_split.inc"
