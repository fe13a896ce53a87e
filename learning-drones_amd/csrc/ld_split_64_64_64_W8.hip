#define LD_TAG 64_64_64
#define LD_WIDTHS 64, 64, 64
#define LD_SPLIT_ONLY_CASE 8
#include "ld_split.inc"
