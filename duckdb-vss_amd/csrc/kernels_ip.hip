// kernels_ip.hip — kernel instantiations for metric ip (see kernels_metric.inc)
#define VSS_MT 2
#include "kernels_metric.inc"
