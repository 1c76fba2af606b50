// kernels_l2sq.hip — kernel instantiations for metric l2sq (see kernels_metric.inc)
#define VSS_MT 0
#include "kernels_metric.inc"
