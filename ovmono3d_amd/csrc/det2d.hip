// RPN + box head device path. (placeholder implementation - filled in after the oracle-2D path is green)
#include "det2d.hpp"
#include <cstring>
namespace ovm {
int det2d_alloc(Det2dWorkspace* w, int, int, int, int, int, int, int, int, std::vector<void*>*) { memset(w, 0, sizeof(*w)); return OVM_OK; }
int det2d_forward(const Det2dModel&, Det2dWorkspace&, float*, float*, int*, int*, float*, int*, hipStream_t) { return OVM_ERR_INVALID; }
int launch_nms_single(const float*, const float*, int, float, int*, int*, hipStream_t) { return OVM_ERR_INVALID; }
}
