// Build-recipe stub (NOT reference source): the reference wraps its counting phases in
// TIME_TRACE_* (utils/perf/timetracer.hpp), which call into ext/src/llvm's TimeProfiler.
// Profiling is off by default (spades/main.cpp:23-48 enables it only on request), so this
// recipe links a disabled profiler instead of compiling the vendored LLVM Support library.
#include <llvm/Support/TimeProfiler.h>
namespace llvm {
TimeTraceProfiler *getTimeTraceProfilerInstance() { return nullptr; }
void timeTraceProfilerBegin(StringRef, StringRef) {}
void timeTraceProfilerEnd() {}
}
