// launchers.h — host-side launch functions of the templated kernels.  The kernels are instantiated per index metric
// in three translation units (kernels_l2sq.hip / kernels_cosine.hip / kernels_ip.hip, all built from
// kernels_metric.inc) so they compile in parallel; the engine picks the instantiation at run time.
#pragma once
#include "exact_kernels.h"
#include "hnsw_kernels.h"

namespace vss {


#ifndef VSS_TEAM_WAVES_N
#define VSS_TEAM_WAVES_N 8 // (4 and 16 measured: A/B builds)
#endif
constexpr int VSS_TEAM_WAVES = VSS_TEAM_WAVES_N; // waves per query of the solo shape's team variant (hnsw_kernels.h, TeamScorer)

struct LaunchCfg {
	uint32_t nch;  // float4 chunks per lane: V <= nch * G  (1, 3, 6 have unrolled instantiations, others loop)
	uint32_t regs; // registers needed by the candidate list = ceil(limit / 64)
	uint32_t grid;
	uint32_t lds; // dynamic LDS bytes
	uint32_t threads; // search engine: threads per workgroup (64 x (walkers + scoring waves))
	hipStream_t stream;
};

template <int MT>
hipError_t launch_search(const SearchArgs &a, const LaunchCfg &c);
template <int MT>
hipError_t launch_search_solo(const SearchArgs &a, const LaunchCfg &c);
template <int MT>
hipError_t launch_phase_a(const BuildArgs &a, const LaunchCfg &c);
template <int MT>
hipError_t launch_phase_b(const LinkArgs &a, const LaunchCfg &c);
template <int MT>
hipError_t launch_clusters(const ClusterArgs &a, const LaunchCfg &c);
template <int MT>
hipError_t launch_rerank(const RerankArgs &a, const LaunchCfg &c);

#define VSS_DECLARE_METRIC(MT)                                                                                         \
	template <>                                                                                                        \
	hipError_t launch_search<MT>(const SearchArgs &, const LaunchCfg &);                                               \
	template <>                                                                                                        \
	hipError_t launch_search_solo<MT>(const SearchArgs &, const LaunchCfg &);                                          \
	template <>                                                                                                        \
	hipError_t launch_phase_a<MT>(const BuildArgs &, const LaunchCfg &);                                               \
	template <>                                                                                                        \
	hipError_t launch_phase_b<MT>(const LinkArgs &, const LaunchCfg &);                                                \
	template <>                                                                                                        \
	hipError_t launch_clusters<MT>(const ClusterArgs &, const LaunchCfg &);                                            \
	template <>                                                                                                        \
	hipError_t launch_rerank<MT>(const RerankArgs &, const LaunchCfg &);
VSS_DECLARE_METRIC(0)
VSS_DECLARE_METRIC(1)
VSS_DECLARE_METRIC(2)

} // namespace vss
