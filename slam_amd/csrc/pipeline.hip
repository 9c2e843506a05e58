// Batched per-frame pipeline kernels (added incrementally).
#include "common.h"
