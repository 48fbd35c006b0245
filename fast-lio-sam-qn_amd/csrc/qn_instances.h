// qn_instances.h - the heavy template kernels (every one inlines the cooperative grid search of qn_device.cuh) are
// explicitly instantiated in separate translation units so that hipcc compiles them in parallel: qn_inst.hip is
// compiled once per group with -DQN_INST_GROUP=<g>; everywhere else (QN_INST_GROUP undefined / 0) the same list
// is a set of explicit-instantiation DECLARATIONS, so the host TU launches the kernels without compiling them.
// Group 1 (this file): histogram k-NN, 1-NN searches, tracking, the optimiser tick (k_tick).  Groups 2-9: qn_instances_knn.h.  Group 10: k_align_persist.  Group 11: the batched forms (k_lanes<F>).
#pragma once
#include "qn_instances_knn.h"
#include "qn_gicp_kernels.cuh"
#include "qn_tick.cuh"
#include "qn_persist.cuh"

#if QN_INST_GROUP == 1
#define QN_G1 template
#else
#define QN_G1 extern template
#endif
#if QN_INST_GROUP == 11        // the batched (pair-as-grid-dimension) forms of the heavy kernels: k_lanes<F>
#define QN_G11 template
#else
#define QN_G11 extern template
#endif
#if QN_INST_GROUP == 10        // the persistent align kernel (qn_persist.cuh): a unit of its own
#define QN_G10 template
#else
#define QN_G10 extern template
#endif

namespace qn {

#define QN_NN_BLOCK 256        // threads per block of the grid-form 1-NN passes (16 queries per wave)
#define QN_KNN_HIST_ARGS (GridView, int, float, int, int32_t*, float*, uint2*, uint32_t*, uint2*, uint32_t*)
#define QN_NN_SEARCH_ARGS (GridView, GridView, const GicpState*, double, float, int, int32_t*, float*, int32_t*, float4*, uint2*, uint32_t*, uint2*, uint32_t*, int, float, uint32_t*, NnOpt)
#define QN_NN_TRACK_ARGS (GridView, GridView, const float4*, const GicpState*, double, int32_t*, float*, int32_t*, float4*, uint2*, uint32_t*, uint2*, uint32_t*)

QN_G1 __global__ void k_knn_hist<false, 32> QN_KNN_HIST_ARGS;
QN_G1 __global__ void k_knn_hist<true, 32> QN_KNN_HIST_ARGS;
QN_G1 __global__ void k_knn_hist<false, 48> QN_KNN_HIST_ARGS;
QN_G1 __global__ void k_knn_hist<false, 32, false> QN_KNN_HIST_ARGS;      // VALU scoring (knob knn_mm 0)
QN_G1 __global__ void k_knn_hist<true, 48> QN_KNN_HIST_ARGS;
QN_G1 __global__ void k_nn_search<0, false, QN_NN_BLOCK> QN_NN_SEARCH_ARGS;
QN_G1 __global__ void k_nn_search<0, true, QN_BLOCK> QN_NN_SEARCH_ARGS;
QN_G1 __global__ void k_nn_search<0, true, QN_BLOCK, true> QN_NN_SEARCH_ARGS;
QN_G1 __global__ void k_nn_search<1, false, QN_NN_BLOCK> QN_NN_SEARCH_ARGS;
QN_G1 __global__ void k_nn_search<1, true, QN_BLOCK> QN_NN_SEARCH_ARGS;
QN_G1 __global__ void k_nn_track<0> QN_NN_TRACK_ARGS;
QN_G1 __global__ void k_nn_track<1> QN_NN_TRACK_ARGS;
QN_G1 __global__ void k_tick<256, 2, 0, false>(TickArgs);
QN_G1 __global__ void k_tick<256, 3, 0, false>(TickArgs);
QN_G1 __global__ void k_tick<256, 4, 0, false>(TickArgs);
QN_G1 __global__ void k_tick<512, 2, 0, false>(TickArgs);
QN_G1 __global__ void k_tick<512, 3, 0, false>(TickArgs);
QN_G1 __global__ void k_tick<512, 4, 0, false>(TickArgs);
QN_G1 __global__ void k_tick<256, 4, 1, false>(TickArgs);
QN_G1 __global__ void k_tick<512, 4, 1, false>(TickArgs);
QN_G1 __global__ void k_tick<256, 4, 0, true>(TickArgs);      // developer variants with device-clock stamps (knob clk_probe)
QN_G1 __global__ void k_tick<512, 4, 0, true>(TickArgs);
// batched launches (k_lanes: blockIdx.y = table entry = one candidate pair / one of its clouds)
QN_G11 __global__ void k_lanes<KnnHistK<false, 32>>(const LaneEntry<KnnHistArgs>*);
QN_G11 __global__ void k_lanes<KnnHistK<true, 32>>(const LaneEntry<KnnHistArgs>*);
QN_G11 __global__ void k_lanes<KnnHistK<false, 48>>(const LaneEntry<KnnHistArgs>*);
QN_G11 __global__ void k_lanes<KnnHistK<false, 32, false>>(const LaneEntry<KnnHistArgs>*);      // VALU scoring (knob knn_mm 0)
QN_G11 __global__ void k_lanes<KnnHistK<true, 48>>(const LaneEntry<KnnHistArgs>*);
QN_G11 __global__ void k_lanes<NnSearchK<0, false, QN_NN_BLOCK>>(const LaneEntry<NnSearchArgs>*);
QN_G11 __global__ void k_lanes<NnSearchK<0, true, QN_BLOCK>>(const LaneEntry<NnSearchArgs>*);
QN_G11 __global__ void k_lanes<NnSearchK<0, true, QN_BLOCK, true>>(const LaneEntry<NnSearchArgs>*);
QN_G11 __global__ void k_lanes<TickK<512, 4, 0, false>>(const LaneEntry<TickArgs>*);
QN_G11 __global__ void k_lanes<TickK<512, 4, 1, false>>(const LaneEntry<TickArgs>*);
QN_G10 __global__ void k_align_persist<QN_PERSIST_TB, false>(PersistArgs);
QN_G10 __global__ void k_align_persist<QN_PERSIST_TB, true>(PersistArgs);      // developer variant with wall-clock stamps (knob persist_probe)

}  // namespace qn
