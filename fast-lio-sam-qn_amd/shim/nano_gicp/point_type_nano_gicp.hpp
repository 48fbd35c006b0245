// Drop-in for <nano_gicp/point_type_nano_gicp.hpp> (included at fast_lio_sam_qn/include/loop_closure.h:16).
// The reference's glue uses pcl::PointXYZI (include/utilities.hpp:36); nothing else is needed here.
#pragma once
#include <pcl/point_types.h>
namespace nano_gicp { using PointType = pcl::PointXYZI; }
