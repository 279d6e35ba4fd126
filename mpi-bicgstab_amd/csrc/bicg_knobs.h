// bicg_knobs.h -- the library's run-time switches come in two kinds.
//   getenv("BICG_...") directly: switches for users and for the tests (tabled in INTEGRATION.md section 6).
//   knob_x("BICG_..."):          measurement knobs of the development rounds -- A/B settings whose outcome is on record in
//                                profiles/NOTES.md, negative results kept for reference. They are read only by a library built
//                                with `make EXPERIMENTS=1` (-DBICG_EXPERIMENTS); the default build has their defaults compiled in
//                                and does not contain the kernels only they can select (k_spmv_sell_fw, k_spmm_dir).
#pragma once

#include <cstdlib>

namespace bicg {

#ifdef BICG_EXPERIMENTS
inline const char *knob_x(const char *name) { return getenv(name); }
constexpr bool kExperiments = true;
#else
inline const char *knob_x(const char *) { return nullptr; }
constexpr bool kExperiments = false;
#endif

}  // namespace bicg
