// mc_univ_kernel instantiations with the dictionaries of the enthalpy pass in LDS (mc_univ.h, UParams::dict_lds)
#include "mc_univ.h"

univ_kernel_fn smolmc_univ_kernel_dict(int sel) { return univ_select<true>(sel); }
