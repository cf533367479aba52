// host_options.hpp — idkptSetDeveloperOption.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

// idkptSetDeveloperOption: tuning / test hooks (DevOptions above).  Unknown names are an error, so that a typo cannot silently test nothing.
static int32_t dev_SetOption(dev_ctx* ctx, const char* name, int32_t value)
{
    if (!ctx || !name) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    DevOptions& o = ctx->opt;
    const std::string n(name);
    if (n == "force_generic") o.forceGeneric = value != 0;
    else if (n == "no_tile_cull") o.noTileCull = value != 0;
    else if (n == "no_lean_primary") o.noLeanPrimary = value != 0;
    else if (n == "leaf_min") o.leafMin = std::max(0, value);
    else if (n == "grab_unit_log2") o.grabUnitLog2 = value;
    else if (n == "grab_fixed") o.grabFixed = value;
    else if (n == "lds_pad") o.ldsPad = value;
    else if (n == "trace_waves") o.traceWaves = std::max(0, value);
    else if (n == "grid_hint") o.gridHint = std::max(0, value);
    else if (n == "grid_rays_x4") o.gridRaysX4 = std::max(0, value);
    else if (n == "grid_mid_waves") o.gridMidWaves = std::max(0, value);
    else if (n == "defer_last") o.deferLast = value != 0;
    else if (n == "split") { REQUIRE(value >= 0 && value <= 3, "idkptSetDeveloperOption: split is 0..3"); o.split = value; }
    else if (n == "split_donor") o.splitDonor = value != 0;
    else if (n == "split_peek") o.splitPeek = std::max(1, value);
    else if (n == "split_scatter") o.splitScatter = std::min(6, std::max(0, value));
    else if (n == "query_scheduler") o.queryScheduler = value != 0;
    else if (n == "group_threads") o.groupThreads = value;
    else if (n == "fused") { REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: fused is 0..2"); o.fused = value; }
    else if (n == "fused_shade_min") o.fusedShadeMin = std::min(64, std::max(1, value));
    else if (n == "leaf_pool") { REQUIRE(value >= -1 && value <= 7, "idkptSetDeveloperOption: leaf_pool is -1 (automatic) or a mask 0..7 (1: primary launches, 2: the first bounce, 4: later bounces)"); o.leafPool = value; }
    else if (n == "pool_min") o.poolMin = std::max(0, value);
    else if (n == "adv_min") o.advMin = std::min(64, std::max(0, value));
#ifdef IDKPT_DEVELOPER
    else if (n == "graph_probe") o.graphProbe = std::max(0, value);
#endif
    else if (n == "inst_tlas_overlap") { REQUIRE(value >= 0 && value <= 100, "idkptSetDeveloperOption: inst_tlas_overlap is a percentage"); FLUSH(); o.instTlasOverlap = value; ctx->itlasValid = false; }
    else if (n == "bounce_pixel_major") { REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: bounce_pixel_major is 0..2"); o.bouncePixelMajor = value; }
    else if (n == "gen_group_max") { REQUIRE(value >= 1 && value <= 16, "idkptSetDeveloperOption: gen_group_max is 1..16"); o.genGroupMax = value; }
    else if (n == "gen_pixel_major") { REQUIRE(value >= 0, "idkptSetDeveloperOption: gen_pixel_major is >= 0"); o.genPixelMajor = value; }
    else if (n == "inst_sieve_overlap") { REQUIRE(value >= 0 && value <= 100, "idkptSetDeveloperOption: inst_sieve_overlap is a percentage"); FLUSH(); o.instSieveOverlap = value; ctx->itlasValid = false; }
    else if (n == "inst_sieve") { REQUIRE(value >= 0, "idkptSetDeveloperOption: inst_sieve is >= 0"); FLUSH(); o.instSieve = value; ctx->itlasValid = false; }
    else if (n == "inst_braid") { REQUIRE(value >= 0, "idkptSetDeveloperOption: inst_braid is >= 0"); FLUSH(); o.instBraid = value; ctx->itlasValid = false; }   // (entries of the own TLAS after partial re-braiding, k_braid; 0: whole instances)
    else if (n == "pair_nodes") { REQUIRE(value == 0 || value == 1, "idkptSetDeveloperOption: pair_nodes is 0 or 1"); o.pairNodes = value; }
    else if (n == "inst_general") { REQUIRE(value >= 0, "idkptSetDeveloperOption: inst_general is >= 0"); FLUSH(); o.instGeneral = value; ctx->itlasValid = false; }
    else if (n == "uni_refill") { REQUIRE(value == 16 || value == 32, "idkptSetDeveloperOption: uni_refill is 16 or 32"); o.uniRefill = value; }
    else if (n == "inst_unify_radius") { REQUIRE(value >= 1 && value <= 512, "idkptSetDeveloperOption: inst_unify_radius is 1-512"); FLUSH(); o.instUnifyRadius = value; ctx->itlasValid = false; }
    else if (n == "inst_unify") { REQUIRE(value >= 0, "idkptSetDeveloperOption: inst_unify is >= 0"); FLUSH(); o.instUnify = value; ctx->itlasValid = false; }   // (subtrees under the unified tree's top, k_unify_*; 0: off)
    else if (n == "inst_tlas") { REQUIRE(value >= 0, "idkptSetDeveloperOption: inst_tlas is >= 0"); FLUSH(); o.instTlas = value; ctx->itlasValid = false; }   // (0: the exact instance loop only; n: the library's own TLAS from n instances on)
    else if (n == "packet") { REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: packet is 0..2"); o.packet = value; ctx->pkState = 0; ctx->pkBatchesSinceProbe = 0; }
    else if (n == "packet_min_live") { REQUIRE(value >= 0 && value <= 100, "idkptSetDeveloperOption: packet_min_live is a percentage"); o.packetMinLive = value; ctx->pkState = 0; ctx->pkBatchesSinceProbe = 0; }
    else if (n == "packet_waves") { REQUIRE(value >= 0 && value <= 32, "idkptSetDeveloperOption: packet_waves is 0 (default) or 1..32"); o.packetWaves = value; }
    else if (n == "wide") { REQUIRE(value >= 0 && value <= 1, "idkptSetDeveloperOption: wide is 0 or 1"); FLUSH(); o.wide = value; }
    else if (n == "wide_cap") { REQUIRE(value >= 0 && value <= 96, "idkptSetDeveloperOption: wide_cap is 0 (default) or 4..96 rows"); o.wideCap = value; }
    else if (n == "wide_count") o.wideCount = value != 0;
    else if (n == "bvh_timing") o.bvhTiming = value != 0;
    else if (n == "bvh_small") o.bvhSmall = value;
    else if (n == "bvh_stackopt_host") o.bvhStackOptHost = value != 0;
    else if (n == "force_no_peer") { }                 // (multi-device contexts: idkpt_api.hpp; nothing to stage on one device)
    else if (n == "trace_variant") {
#ifdef IDKPT_DEVELOPER
        o.traceVariant = value;
#else
        REQUIRE(value == 0 || value == 100, "idkptSetDeveloperOption: trace_variant needs the developer build of the library (libidkpt_dev.so)");
#endif
    }
    else return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "idkptSetDeveloperOption: unknown option '" + n + "'");
    return IDKPT_OK;
}
