#define LD_TAG 48_48
#define LD_WIDTHS 48, 48
#define LD_SPLIT_ONLY_CASE 1
#include "ld_split.inc"
