// Explicit instantiations of the forward kernels, group A (see dtqn_forward_body.hpp).
#include "dtqn_forward_body.hpp"

namespace dtqn {
DTQN_FWD_GROUP_A(DTQN_FWD_DEF)
}  // namespace dtqn
