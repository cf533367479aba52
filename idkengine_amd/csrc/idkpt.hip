// idkpt.hip — libidkpt.so: kernels' __global__ entry points and the C-ABI of include/idkpt.h.
// Host schedule mirrors PathTracer.Compute (Source/Render/PathTracer.cs:214-271); see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include "../../include/idkpt.h"
#include "pt_kernels.hpp"

using namespace ptd;

#define WAVE 64
#define MAX_DEPTH_SLOTS 64
#define WORK_WORDS (GRAB_SLICES * GRAB_STRIDE)   // work counters: launch j uses word j of every 1-KB line (k_trace2: one line per work-list slice, kernels_trace.hpp)

// =================================================================================================== kernels (one translation unit)
#include "kernels_common.hpp"
#include "kernels_trace.hpp"
#include "kernels_wide.hpp"
#include "kernels_trace_inst.hpp"
#include "kernels_packet.hpp"
#include "kernels_trace_split.hpp"
#include "kernels_query.hpp"
#include "kernels_shade.hpp"
#include "kernels_trace_fused.hpp"
#include "kernels_queue.hpp"
#include "kernels_frame.hpp"
#include "kernels_scene.hpp"
#include "bvh_gpu.hpp"
#include "bvh_gpu_full.hpp"

// =================================================================================================== host side (one file per concern, in dependency order)
#include "host_context.hpp"
#include "host_launch.hpp"
#include "host_frame.hpp"
#include "host_scene.hpp"
#include "host_builder.hpp"
#include "host_queries.hpp"
#include "host_schedule.hpp"
#include "host_readback.hpp"
#include "host_options.hpp"
#include "transport_rccl.hpp"

#include "idkpt_api.hpp"
