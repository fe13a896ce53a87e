#define LD_TAG 32_32_32
#define LD_WIDTHS 32, 32, 32
#define LD_SPLIT_ONLY_CASE 1
#include "ld_split.inc"
