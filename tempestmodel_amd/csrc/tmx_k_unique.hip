// tmx_k_unique.hip -- kernels of the node-unique state layout: conversions between the element-major
// layout with duplicated seam nodes and the per-patch node-unique slabs, and the seam averaging restricted to the nodes
// on tile and patch edges.
#include "tmx_device.h"
