// Instantiations of the fused head / operand kernel for DT in {4, 6, 8} (see prep_kernel.h).
#include "prep_kernel.h"

namespace pilco {

void launch_prep_4(const PrepLaunch& a) { launch_prep_dt<4>(a); }
void launch_prep_6(const PrepLaunch& a) { launch_prep_dt<6>(a); }
void launch_prep_8(const PrepLaunch& a) { launch_prep_dt<8>(a); }

}  // namespace pilco
