// qn_instances_knn.h - explicit instantiations (group g) / declarations (every other unit) of the sorted-list k-NN kernel
// k_knn_cov<KMAX, LIST, 4>.  One unit per instantiation (~15 s each for the fully unrolled BestK<24> merges).  These units include only qn_knn_kernels.cuh + qn_device.cuh (see qn_instances.h).
#pragma once
#include "qn_knn_kernels.cuh"

#ifndef QN_INST_GROUP
#define QN_INST_GROUP 0
#endif
#define QN_NUM_INST_GROUPS 11

#if QN_INST_GROUP == 2
#define QN_G2 template
#else
#define QN_G2 extern template
#endif
#if QN_INST_GROUP == 3
#define QN_G3 template
#else
#define QN_G3 extern template
#endif
#if QN_INST_GROUP == 4
#define QN_G4 template
#else
#define QN_G4 extern template
#endif
#if QN_INST_GROUP == 5
#define QN_G5 template
#else
#define QN_G5 extern template
#endif
#if QN_INST_GROUP == 6
#define QN_G6 template
#else
#define QN_G6 extern template
#endif
#if QN_INST_GROUP == 7
#define QN_G7 template
#else
#define QN_G7 extern template
#endif
#if QN_INST_GROUP == 8
#define QN_G8 template
#else
#define QN_G8 extern template
#endif
#if QN_INST_GROUP == 9
#define QN_G9 template
#else
#define QN_G9 extern template
#endif

namespace qn {
#define QN_KNN_COV_ARGS (GridView, const float4*, int, float, int, double*, int32_t*, float*, uint2*, uint32_t*)
// groups 2-5: list form (the exact tail of both k-NN paths); groups 6-9: grid form (knn_hist = 0)
QN_G2 __global__ void k_knn_cov<16, true, 4> QN_KNN_COV_ARGS;
QN_G3 __global__ void k_knn_cov<20, true, 4> QN_KNN_COV_ARGS;
QN_G4 __global__ void k_knn_cov<24, true, 4> QN_KNN_COV_ARGS;
QN_G5 __global__ void k_knn_cov<32, true, 4> QN_KNN_COV_ARGS;
QN_G2 __global__ void k_lanes<KnnCovK<16, true, 4>>(const LaneEntry<KnnCovArgs>*);      // batched forms of the exact tail (k_lanes: blockIdx.y = one cloud of one candidate pair)
QN_G3 __global__ void k_lanes<KnnCovK<20, true, 4>>(const LaneEntry<KnnCovArgs>*);
QN_G4 __global__ void k_lanes<KnnCovK<24, true, 4>>(const LaneEntry<KnnCovArgs>*);
QN_G5 __global__ void k_lanes<KnnCovK<32, true, 4>>(const LaneEntry<KnnCovArgs>*);
QN_G6 __global__ void k_knn_cov<16, false, 4> QN_KNN_COV_ARGS;
QN_G7 __global__ void k_knn_cov<20, false, 4> QN_KNN_COV_ARGS;
QN_G8 __global__ void k_knn_cov<24, false, 4> QN_KNN_COV_ARGS;
QN_G9 __global__ void k_knn_cov<32, false, 4> QN_KNN_COV_ARGS;
}  // namespace qn
