// Instantiations of the fused head / operand kernel for DT in {32} (see prep_kernel.h).
#include "prep_kernel.h"

namespace pilco {

void launch_prep_32(const PrepLaunch& a) { launch_prep_dt<32>(a); }

}  // namespace pilco
