// libbadger_pf_hip.so -- host engine and C-ABI (include/badger_pf.h) for the MI355X
// sensor-update + resample path.  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/badger_pf.h"
#include "device_types.hpp"
#include "kdhist.hpp"
#include "kernels_cloud.hpp"
#include "kernels_lut3d.hpp"
#include "kernels_motion.hpp"
#include "kernels_kld.hpp"
#include "kernels_kld2.hpp"
#include "kernels_recovery.hpp"
#include "kernels_pf.hpp"
#include "kernels_fused.hpp"
#include "kernels_stats.hpp"
#include "kernels_score.hpp"
#include "kernels_window.hpp"

using namespace bpf;

#include "engine_state.hpp"

namespace
{
#include "host_common.inl"
#include "host_scoring.inl"
#include "host_resample.inl"
}  // namespace

// ====================================================================== C-ABI (include/badger_pf.h)
extern "C" {
namespace
{
void mailbox_release(bpf_engine* e);     // abi_mailbox.inl
void collective_release(bpf_engine* e);  // abi_bootstrap.inl
void host_buffers_release(bpf_engine* e);  // abi_hostbuf.inl
}
#include "abi_lifecycle.inl"
#include "abi_hostbuf.inl"
#include "abi_map2d.inl"
#include "abi_planar.inl"
#include "abi_filter.inl"
#include "abi_motion.inl"
#include "abi_statistics.inl"
#include "abi_lut_reference.inl"
#include "abi_wire.inl"
#include "abi_cloud3d.inl"
#include "abi_mailbox.inl"
#include "abi_sharded.inl"
#include "abi_mailbox_step.inl"
#include "abi_bootstrap.inl"
#include "abi_measure.inl"
}  // extern "C"
