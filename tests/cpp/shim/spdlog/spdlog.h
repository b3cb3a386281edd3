// Minimal stand-in for spdlog (tests/cpp/shim/README.md): every level from info up prints the raw format string to stderr, followed by
// the string-like and numeric arguments (enough to see WHICH profile label a "[profile] {:22s} ..." line of include/Profiling.h carries).
#pragma once
#include <cstdio>
#include <memory>
#include <string>
#include <type_traits>
namespace spdlog {
namespace detail {
template <class T> void put_arg(const T& v) {
  if constexpr (std::is_convertible<T, std::string>::value) std::fprintf(stderr, " | %s", std::string(v).c_str());
  else if constexpr (std::is_arithmetic<T>::value) std::fprintf(stderr, " | %g", static_cast<double>(v));
}
}  // namespace detail
class logger {
public:
  template <class... A> void trace(const std::string&, A&&...) {}
  template <class... A> void debug(const std::string&, A&&...) {}
  template <class... A> void info(const std::string& f, A&&... a) {
    std::fprintf(stderr, "[info] %s", f.c_str());
    (detail::put_arg(a), ...);
    std::fprintf(stderr, "\n");
  }
  template <class... A> void warn(const std::string& f, A&&... a) { put("[warn] ", f, a...); }
  template <class... A> void error(const std::string& f, A&&... a) { put("[error] ", f, a...); }
  template <class... A> void critical(const std::string& f, A&&... a) { put("[critical] ", f, a...); }

private:
  template <class... A> static void put(const char* lvl, const std::string& f, const A&... a) {
    std::fprintf(stderr, "%s%s", lvl, f.c_str());
    (detail::put_arg(a), ...);
    std::fprintf(stderr, "\n");
  }

public:
};
}  // namespace spdlog
