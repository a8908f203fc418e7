// isa_scatter.hip -- compile-only harness: instantiates the shipped shapes of the two scatter kernels so that their ISA and
// resource usage can be inspected in seconds instead of recompiling render_fused.hip (2.5 min).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I include -I <pkg>/csrc -c tools/isa_scatter.hip \
//         -save-temps=obj -Rpass-analysis=kernel-resource-usage -o /tmp/isa/s.o
#include "hash_kernels.h"
#include "scatter_binned.h"

namespace naf {
template __global__ void scatter_bin_kernel<BF16, 2, SrcRays, PairBF16<2>, 512, 2, 16>(SrcRays, const uint16_t *, const int32_t *, float *, PairBF16<2> *, uint32_t *, uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, BinPlan, SlabReduce, uint32_t, uint32_t);
template __global__ void scatter_bin_kernel<BF16, 2, SrcRays, PairBF16<2>, 512, 2, 4>(SrcRays, const uint16_t *, const int32_t *, float *, PairBF16<2> *, uint32_t *, uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, BinPlan, SlabReduce, uint32_t, uint32_t);
template __global__ void scatter_reduce_kernel<2, PairBF16<2>, true>(const PairBF16<2> *, const uint32_t *, const int32_t *, float *, const uint32_t *, uint32_t, uint32_t, uint32_t, BinPlan, AdamTail);
template __global__ void scatter_reduce_kernel<2, PairBF16<2>, false>(const PairBF16<2> *, const uint32_t *, const int32_t *, float *, const uint32_t *, uint32_t, uint32_t, uint32_t, BinPlan, AdamTail);
}
#include "scatter_v2.h"
namespace naf {
template __global__ void scatter_bin2_kernel<512, 16, 6>(SrcRays, const uint16_t *, const int32_t *, float *, PairFx *, uint32_t *, uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, BinPlan, SlabReduce, DrawJob);
template __global__ void scatter_bin2_kernel<1024, 16, 0>(SrcRays, const uint16_t *, const int32_t *, float *, PairFx *, uint32_t *, uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, BinPlan, SlabReduce, DrawJob);
}
namespace naf {
template __global__ void scatter_reduce2_kernel<true, true>(const PairFx *, const uint32_t *, const int32_t *, float *, const uint32_t *, uint32_t, uint32_t, uint32_t, BinPlan, AdamTail);
template __global__ void scatter_reduce2_kernel<false, false>(const PairFx *, const uint32_t *, const int32_t *, float *, const uint32_t *, uint32_t, uint32_t, uint32_t, BinPlan, AdamTail);
}
