// msda_env.h -- every environment variable the library reads, in ONE table.
//
// They are tuning and test hooks (which formulation runs, how a launch is cut); none changes a result.  The table is
// read once, at the first use, and again only when mmfs_env_reload() is called (the tests flip knobs inside one
// process and say so); no launch path reads the environment itself.  include/mmfs_msda.h: mmfs_env_reload, mmfs_env_knob.
#pragma once

namespace mmfs {

enum Knob {
    K_FWD_ALGO, K_FWD_MMA_LDS_KB, K_FWD_MMA_QPW, K_FWD_Q8_LDS_KB, K_FWD_Q8_QPR, K_FWD_WQ_LDS_KB, K_FWD_WQ_QPW,
    K_TAPS_ALGO, K_TAPS_MMA_QPW, K_MMA_GRID, K_MMA_PERSIST, K_HYBRID, K_DOT_CHUNKS,
    K_VALUE_ALGO, K_PREPARE_IN_TAPS,
    K_NT_MIN, K_SORT_WINDOW_KB, K_SORT_ROUNDS, K_SORT_SMALL, K_SORT_MANY_POINTS, K_SORT_HGROUP, K_SORT_REPACK, K_SORT_WIDE,
    K_SAMPLE_DECODE, K_LIN_ROWS, K_LIN_UNROLL, K_LIN_EARLY, K_QUERY_LDS_KB, K_NORM_BWD_GRID,
    K_COUNT
};

struct KnobInfo { const char *name, *doc; };
const KnobInfo &knob_info(int k);

// the variable's value as it was when the table was read: nullptr when unset
const char *knob_str(Knob k);
// atoi of the value; ``dflt`` when unset or empty
int knob_int(Knob k, int dflt);
long long knob_ll(Knob k, long long dflt);

}  // namespace mmfs
