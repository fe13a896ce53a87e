#define LD_TAG 64_64_64
#define LD_WIDTHS 64, 64, 64
#define LD_ONLY_F32
#include "ld_unrolled.inc"
