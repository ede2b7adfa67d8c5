// mtr_knobs.h — experiment knobs.  The RELEASE library never reads the environment: a stray variable must not change
// a user's performance or results.  Only a build with -DMTR_EXPERIMENTS (make experiments -> libmitransient_amd_exp.so; the
// test-only host harness) honours MTR_* variables; tools/ uses that build (MTR_LIB=...).
#ifndef MTR_KNOBS_H
#define MTR_KNOBS_H
#include <cstdlib>
namespace mtr {
#ifdef MTR_EXPERIMENTS
inline const char *knob(const char *name) { return std::getenv(name); }
#else
inline const char *knob(const char *) { return nullptr; }
#endif
}
#endif
