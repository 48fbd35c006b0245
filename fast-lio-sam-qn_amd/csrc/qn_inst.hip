// qn_inst.hip - explicit instantiations of the heavy search kernels; compiled once per group (qn_instances.h,
// qn_instances_knn.h).  Groups >= 2 see nothing but the sorted-list k-NN kernel, so edits elsewhere do not rebuild them.
#include <hip/hip_runtime.h>
#if !defined(QN_INST_GROUP) || QN_INST_GROUP < 1
#error "compile with -DQN_INST_GROUP=<1..QN_NUM_INST_GROUPS>"
#endif
#if QN_INST_GROUP == 1 || QN_INST_GROUP == 10 || QN_INST_GROUP == 11
#include "qn_instances.h"
#else
#include "qn_instances_knn.h"
#endif
