// Instantiations of the fused head / operand kernel for DT in {10, 11} (see prep_kernel.h).
#include "prep_kernel.h"

namespace pilco {

void launch_prep_10(const PrepLaunch& a) { launch_prep_dt<10>(a); }
void launch_prep_11(const PrepLaunch& a) { launch_prep_dt<11>(a); }

}  // namespace pilco
