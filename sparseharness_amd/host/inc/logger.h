// logger.h -- levelled stderr logger for the host mirror.
// Stands in for the reference's executor::Logger (inc/util/Logger.h:63-92),
// which cannot be reused because it is typed on cl::Error / CL/cl.hpp.
// Same macro names and the same "[file:line ss.mmms LEVEL]" header shape
// (src/util/Logger.cpp:226-251); variadic arguments are streamed in order.
#pragma once
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace shlog {
enum Level { Error = 1, Warning, Info, Debug, DebugInfo };
inline Level threshold() {
  static Level lvl = [] {
    const char *e = std::getenv("SH_LOG_LEVEL"); // 1..5, default Info
    int v = e ? std::atoi(e) : 3;
    return (Level)(v < 1 ? 1 : v > 5 ? 5 : v);
  }();
  return lvl;
}
inline std::chrono::steady_clock::time_point t0() {
  static auto t = std::chrono::steady_clock::now();
  return t;
}
inline void put(std::ostream &) {}
template <typename A, typename... R> void put(std::ostream &o, A &&a, R &&...r) {
  o << a;
  put(o, std::forward<R>(r)...);
}
template <typename... Args> void log(Level lvl, const char *tag, const char *file, int line, Args &&...args) {
  if (lvl > threshold())
    return;
  auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0()).count();
  const char *base = std::strrchr(file, '/');
  std::ostringstream o;
  o << "[" << (base ? base + 1 : file) << ":" << line << " " << ms / 1000 << "." << std::setw(3)
    << std::setfill('0') << ms % 1000 << "s " << tag << "] ";
  put(o, std::forward<Args>(args)...);
  o << "\n";
  std::clog << o.str();
}
} // namespace shlog

#define LOG_ERROR(...) shlog::log(shlog::Error, "ERROR", __FILE__, __LINE__, __VA_ARGS__)
#define LOG_WARNING(...) shlog::log(shlog::Warning, "WARNING", __FILE__, __LINE__, __VA_ARGS__)
#define LOG_INFO(...) shlog::log(shlog::Info, "INFO", __FILE__, __LINE__, __VA_ARGS__)
#define LOG_DEBUG(...) shlog::log(shlog::Debug, "DEBUG", __FILE__, __LINE__, __VA_ARGS__)
#define LOG_DEBUG_INFO(...) shlog::log(shlog::DebugInfo, "DINFO", __FILE__, __LINE__, __VA_ARGS__)
