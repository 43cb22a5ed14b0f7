// Explicit instantiations of the forward kernels, group E (see dtqn_forward_body.hpp).
#include "dtqn_forward_body.hpp"

namespace dtqn {
DTQN_FWD_GROUP_E(DTQN_FWDL_DEF)
}  // namespace dtqn
