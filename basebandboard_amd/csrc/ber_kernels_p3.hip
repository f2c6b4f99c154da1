// ber_kernels_p3.hip -- more instances of the fused trial kernel (see ber_kernels.hip / ber_kernels_impl.hpp): a unit of its own so that
// make -j compiles them side by side.
#define BBB_BER_PART 3
#include "ber_kernels.hip"
