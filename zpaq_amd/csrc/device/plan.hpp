#pragma once
#include <string>
#include <vector>

#include "../host/common.hpp"
#include "layout.h"

struct zpq_plan {
  std::vector<uint8_t> header;     // stored header bytes
  std::vector<uint8_t> blob;       // PlanHeader + CompDesc[] + Segment[] + prog (host copy)
  double memory = 0;               // ZPAQL::memory()
  double algo_bytes = 0;
  // Everything a plan owns ON A DEVICE -- its uploaded copy and the loaded code objects -- exists once per GPU the
  // engine drives (one engine per device, host-buffer batches are sharded over them).  The loaders and the engine
  // address the slot of the device the calling thread works for (zpq::plan_device_index()).
  struct OnDevice {
    void* d_blob = nullptr;          // device copy (lazily uploaded by the engine)
    // per-header specialised kernels (device/spec_loader.hpp), one per workgroup shape: [0] 4 blocks per workgroup
    // (all side tables in LDS, one wavefront per SIMD), [1] 8 (two per SIMD); the engine chooses per launch.
    // (12/16-block shapes were measured in round 2 and lost: profiles/r02_ab_matrix.txt.)
    void* spec[4] = {nullptr, nullptr, nullptr, nullptr};   // SpecKernel*: 4 / 8 blocks per workgroup, two blocks per wavefront (decoder), lockstep decoder
    int spec_state[4] = {0, 0, 0, 0};              // 0 not tried, 1 loaded, -1 unavailable
    // PipeKernel*: the pipelined encoder (device/pipe_kernel.h) in its variants (host/codegen.hpp pipe_options):
    // [0] throughput shape, [1] latency shape, [2] latency shape with 2048-byte steps
    void* pipe[4] = {nullptr, nullptr, nullptr, nullptr};      // per encoder variant (host/codegen.hpp pipe_options)
    int pipe_state[4] = {0, 0, 0, 0};
    std::string pipe_note, spec_note;     // where the last kernel came from / why it is unavailable
  };
  static const int kMaxDevices = 16;
  OnDevice dev[kMaxDevices];
  OnDevice& cur();
  const OnDevice& cur() const;
  const zpq::PlanHeader& hdr() const { return *(const zpq::PlanHeader*)blob.data(); }
  const zpq::CompDesc* comps() const { return (const zpq::CompDesc*)(blob.data() + hdr().off_comp); }
};

namespace zpq {
// device slot the calling thread works for (set by the engine around every use of a plan's device state)
int plan_device_index();
void set_plan_device_index(int dev);
// Parses a stored block header into a plan; throws Failure(ZPQ_E_HEADER/...).
zpq_plan* plan_from_header(const U8* header, size_t hlen, bool list_only = false);
}

inline zpq_plan::OnDevice& zpq_plan::cur() { return dev[zpq::plan_device_index()]; }
inline const zpq_plan::OnDevice& zpq_plan::cur() const { return dev[zpq::plan_device_index()]; }
