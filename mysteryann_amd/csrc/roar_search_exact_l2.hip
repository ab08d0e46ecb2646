// roar_search_exact_l2.hip -- SearchRoarGraph kernels, visited mode EXACT (u8 epoch tags in HBM), metric L2
// (one translation unit per pair so that the build runs in parallel; the code is roar_search_v2.h).
#define ROAR_TU_MODE 0
#define ROAR_TU_IP 0
#include "roar_search_v2.h"
