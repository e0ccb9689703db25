// msda_env.hip -- the one table of environment knobs (msda_env.h) and its C-ABI face.
#include "msda_env.h"
#include "../../include/mmfs_msda.h"
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <string>

namespace mmfs {
namespace {

const KnobInfo kKnobs[K_COUNT] = {
    {"MMFS_FWD_ALGO", "forward formulation wherever it is supported: vec (row gather) | mma (LDS levels) | q8 (32-channel slices) | wq (a wave per query)"},
    {"MMFS_FWD_MMA_LDS_KB", "msda_fwd_mma: LDS budget of a workgroup (tests: forces levels out of LDS)"},
    {"MMFS_FWD_MMA_QPW", "msda_fwd_mma: queries per run"},
    {"MMFS_FWD_Q8_LDS_KB", "msda_fwd_q8: LDS budget of a workgroup"},
    {"MMFS_FWD_Q8_QPR", "msda_fwd_q8: queries per run"},
    {"MMFS_FWD_WQ_LDS_KB", "msda_fwd_wq: LDS budget of a workgroup"},
    {"MMFS_FWD_WQ_QPW", "msda_fwd_wq: queries per run"},
    {"MMFS_TAPS_ALGO", "grad_loc / grad_attn: vec (row gather) | mma (LDS levels on the matrix cores); either one also refuses mmfs_msda_backward_sorted"},
    {"MMFS_TAPS_MMA_QPW", "msda_taps_mma: queries per run"},
    {"MMFS_MMA_GRID", "persistent kernels: this many workgroups whatever the shape"},
    {"MMFS_MMA_PERSIST", "0: one workgroup per run of queries instead of one per CU"},
    {"MMFS_HYBRID", "0: no dense small-level product for grad_loc / grad_attn (row gather for every level)"},
    {"MMFS_DOT_CHUNKS", "1: dense grad_loc / grad_attn also for levels of several 256-pixel chunks"},
    {"MMFS_VALUE_ALGO", "grad_value generation: block (2x2-block reduce) | pixel (first generation); default: matrix-core tile reduce"},
    {"MMFS_PREPARE_IN_TAPS", "0: the grad_value half's opening job is a launch of its own, not hosted by msda_taps_mma"},
    {"MMFS_NT_MIN", "cell sort: tiles per level at least"},
    {"MMFS_SORT_WINDOW_KB", "cell sort: LDS window"},
    {"MMFS_SORT_ROUNDS", "cell sort: most window rounds before the two-scan path (0: always that path)"},
    {"MMFS_SORT_SMALL", "0: always 1024-lane sort workgroups (no 256 / 512-lane variants for many small slices)"},
    {"MMFS_SORT_MANY_POINTS", "0: no grouped scan for more than two 16-byte vectors of locations per query"},
    {"MMFS_SORT_HGROUP", "cell sort: heads that share a workgroup's sector reads"},
    {"MMFS_SORT_REPACK", "1: re-pack loc / attn level-major before the sort even where it could read them in place"},
    {"MMFS_SORT_WIDE", "0: two-vector queries (P = 8, 16-bit) on the 1024-lane sort instead of 512 lanes with eight samples each; 2: one-vector queries on 512 lanes too (measured slower, r06x)"},
    {"MMFS_SAMPLE_DECODE", "0: decode-sized fused-sampler calls on the in-order kernel"},
    {"MMFS_LIN_ROWS", "mmfs_linear_small: token rows per wave (1 | 2)"},
    {"MMFS_LIN_UNROLL", "mmfs_linear_small: weight pieces in flight (4 | 8)"},
    {"MMFS_LIN_EARLY", "mmfs_linear_small: 0: no early issue of the next weight pieces"},
    {"MMFS_QUERY_LDS_KB", "mmfs_query_prep: LDS budget of a workgroup"},
    {"MMFS_NORM_BWD_GRID", "RMS-norm backward: workgroups"},
};

struct Values {
    std::string text[K_COUNT];
    bool set[K_COUNT];
};
// The table the launch paths read is reached through one atomic pointer; a reload publishes a NEW table and never frees the
// old one (a launch on another thread may still hold a pointer into its strings: ADVICE r5 -- reassigning the strings in
// place was a data race).  Reloads are a test / measurement affair: a few hundred bytes each.
std::atomic<const Values *> g_values{nullptr};
std::mutex g_reload;

const Values *read_all()
{
    Values *v = new Values;
    for (int k = 0; k < K_COUNT; ++k) {
        const char *e = std::getenv(kKnobs[k].name);
        v->set[k] = e != nullptr;
        v->text[k] = e ? e : "";
    }
    return v;
}

const Values *current()
{
    const Values *v = g_values.load(std::memory_order_acquire);
    if (v) return v;
    std::lock_guard<std::mutex> lock(g_reload);
    v = g_values.load(std::memory_order_acquire);
    if (!v) {
        v = read_all();
        g_values.store(v, std::memory_order_release);
    }
    return v;
}

}  // namespace

const KnobInfo &knob_info(int k) { return kKnobs[k]; }

const char *knob_str(Knob k)
{
    const Values *v = current();
    return v->set[k] ? v->text[k].c_str() : nullptr;
}

int knob_int(Knob k, int dflt)
{
    const char *e = knob_str(k);
    return (e && *e) ? std::atoi(e) : dflt;
}

long long knob_ll(Knob k, long long dflt)
{
    const char *e = knob_str(k);
    return (e && *e) ? std::atoll(e) : dflt;
}

}  // namespace mmfs

extern "C" void mmfs_env_reload(void)
{
    std::lock_guard<std::mutex> lock(mmfs::g_reload);
    mmfs::g_values.store(mmfs::read_all(), std::memory_order_release);
}

extern "C" int mmfs_env_knob(int index, const char **name, const char **doc, const char **value)
{
    if (index < 0 || index >= mmfs::K_COUNT) return MMFS_E_DIMS;
    if (name) *name = mmfs::knob_info(index).name;
    if (doc) *doc = mmfs::knob_info(index).doc;
    if (value) *value = mmfs::knob_str((mmfs::Knob)index);
    return MMFS_OK;
}
