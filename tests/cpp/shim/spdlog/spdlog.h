// Minimal stand-in for spdlog (tests/cpp/shim/README.md): every level prints the raw format string to stderr.
#pragma once
#include <cstdio>
#include <memory>
#include <string>
namespace spdlog {
class logger {
public:
  template <class... A> void trace(const std::string&, A&&...) {}
  template <class... A> void debug(const std::string&, A&&...) {}
  template <class... A> void info(const std::string& f, A&&...) { std::fprintf(stderr, "[info] %s\n", f.c_str()); }
  template <class... A> void warn(const std::string& f, A&&...) { std::fprintf(stderr, "[warn] %s\n", f.c_str()); }
  template <class... A> void error(const std::string& f, A&&...) { std::fprintf(stderr, "[error] %s\n", f.c_str()); }
  template <class... A> void critical(const std::string& f, A&&...) { std::fprintf(stderr, "[critical] %s\n", f.c_str()); }
};
}  // namespace spdlog
