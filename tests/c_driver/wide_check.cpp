// wide_check.cpp — exposes the host-side build of the wide nodes (idkengine_amd/csrc/wide_nodes.hpp, wide::build_host) to tests/test_gpu_wide.py, which compares
// it byte for byte with what the device kernels (k_wide_topo, k_wide_fill) derive.  Test infrastructure; built by the test with g++ -ffp-contract=off.
#include "../../idkengine_amd/csrc/wide_nodes.hpp"
#include <string.h>

extern "C" {
// returns the number of wide nodes; *leafUnits = 16-byte units of leaf records.  Buffers may be null (size query).
int wide_host_build(const void* nodes, unsigned nodeCount, const float* triVerts, void* outNodes, unsigned capNodes, void* outLeaves, unsigned capLeafUnits, unsigned* leafUnits)
{
    wide::HostBuild B = wide::build_host((const wide::Bvh2Node*)nodes, nodeCount, triVerts);
    *leafUnits = (unsigned)(B.leafRecs.size() / 4);
    if (outNodes && capNodes >= B.nodes.size()) memcpy(outNodes, B.nodes.data(), B.nodes.size() * sizeof(wide::Node));
    if (outLeaves && capLeafUnits >= *leafUnits) memcpy(outLeaves, B.leafRecs.data(), B.leafRecs.size() * 4);
    return (int)B.nodes.size();
}
}
