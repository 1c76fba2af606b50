// kernels_cosine.hip — kernel instantiations for metric cosine (see kernels_metric.inc)
#define VSS_MT 1
#include "kernels_metric.inc"
