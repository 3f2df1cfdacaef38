// csds_timer.h -- RAII scope timers emitting the reference's profiling lines.
// Keeps the exact line formats of the reference so scripts/experiments/analyse.sh
// (:16-43) still parses them:
//   PFTimerStart("name", "ctx") / PROFILING_DATUM("name", "ctx", <ms>, "C++") /
//   PFTimerEnd("name", "ctx")   (reference: src/csds_timer.cpp:30-58)
// Scope timers write to stderr, report_timing (device-side times) to stdout,
// as in the reference (inc/csds_timer.h:10-14, src/csds_timer.cpp:41-48).
// Header-only; SH_NO_TREE_PERF drops the Start/End bracket lines.
#pragma once
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>

#include "common.h"

class CSDSTimer {
public:
  // SH_QUIET_TIMERS=1 silences scope timers (set by the Python binding, which
  // loads this code as a library rather than as an app).
  static bool quiet() {
    static bool q = [] { const char *e = std::getenv("SH_QUIET_TIMERS"); return e && e[0] == '1'; }();
    return q;
  }
  CSDSTimer(const char *name, const char *context, std::ostream &os = std::cerr)
      : _name(name), _ctx(context), _os(&os), _t0(std::chrono::system_clock::now()) {
    if (quiet()) return;
#ifndef SH_NO_TREE_PERF
    *_os << "PFTimerStart(\"" << _name << "\", \"" << _ctx << "\")" << ENDL;
#endif
  }
  ~CSDSTimer() {
    if (quiet()) return;
    auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now() - _t0);
    datum(*_os, _name, _ctx, ns);
#ifndef SH_NO_TREE_PERF
    *_os << "PFTimerEnd(\"" << _name << "\", \"" << _ctx << "\")" << ENDL;
#endif
  }
  static void reportTiming(const std::string &name, const std::string &context, std::chrono::nanoseconds ns) {
    datum(std::cout, name, context, ns);
  }

private:
  static void datum(std::ostream &os, const std::string &n, const std::string &c, std::chrono::nanoseconds ns) {
    os << "PROFILING_DATUM(\"" << n << "\", \"" << c << "\", " << ((double)ns.count()) / 1000000.0 << ", \"C++\")" << ENDL;
  }
  std::string _name, _ctx;
  std::ostream *_os;
  std::chrono::time_point<std::chrono::system_clock> _t0;
};

#define SH_TIMER_CAT2(a, b) a##b
#define SH_TIMER_CAT(a, b) SH_TIMER_CAT2(a, b)
#define start_timer(name, context) CSDSTimer SH_TIMER_CAT(_csds_timer_, __LINE__)(#name, #context, std::cerr);
#define report_timing(name, context, time) CSDSTimer::reportTiming(#name, #context, std::chrono::nanoseconds(time));
