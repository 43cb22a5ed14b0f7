// Explicit instantiations of the forward kernels, group D (see dtqn_forward_body.hpp).
#include "dtqn_forward_body.hpp"

namespace dtqn {
DTQN_FWD_GROUP_D(DTQN_FWD2_DEF)
}  // namespace dtqn
