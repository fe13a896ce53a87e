#define LD_TAG 16
#define LD_WIDTHS 16
#include "ld_split.inc"
