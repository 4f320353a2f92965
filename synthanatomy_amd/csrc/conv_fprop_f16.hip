// f16 forward-operand instances of the implicit-GEMM forward kernels (conv_fprop_kernels.h): the encoder chain of the VQ-VAE runs its forward
// pass on IEEE-half activations and weights -- the reference's AMP dtype (src/engines/trainer.py:161-163, run_vqvae.py --amp), three more
// mantissa bits than bf16 at the same MFMA rate -- so that the code indices of the throughput mode follow the fp32 path; gradients stay bf16.
#include "conv_fprop_kernels.h"

namespace sa {

int dispatch_fprop_f16(const FpropArgs& a, hipStream_t st) { return dispatch_fprop<f16_t>(a, st); }
int launch_resblock_f16(const FpropArgs& a, hipStream_t st) { return launch_resblock<f16_t>(a, st); }

}  // namespace sa
