// evolve/brain.py:10 default layer_spec {1: 8, 2: 10, 3: 4}
#define LD_TAG 8_10
#define LD_WIDTHS 8, 10
#include "ld_unrolled.inc"
