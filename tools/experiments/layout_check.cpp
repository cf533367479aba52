// layout_check.cpp — exposes idkengine_amd/csrc/node_layout.hpp (host side of the derived node order) to tests/test_node_layout.py.
#include "../../idkengine_amd/csrc/node_layout.hpp"
extern "C" int layout_compute(const void* nodes, int nodeCount, unsigned basePair, int mode, int treeletDepth, unsigned* outSlots)
{
    std::vector<uint32_t> slot;
    const bool ok = nodelayout::compute((const nodelayout::Node*)nodes, nodeCount, basePair, mode, treeletDepth, slot);
    for (size_t i = 0; i < slot.size(); i++) outSlots[i] = slot[i];
    return ok ? 1 : 0;
}
