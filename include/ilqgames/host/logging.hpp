// CHECK / LOG / flag macros for the host-side mirror.
//
// The reference reports programmer errors through glog CHECKs (abort with a message) and reads
// example parameters from gflags.  Neither library is a dependency here; this header provides the
// same *spellings* with a few dozen lines so problem-definition code written for the reference
// compiles, and CHECK failures keep the reference's behaviour (message on stderr, abort()).
#ifndef ILQGAMES_HOST_LOGGING_HPP_
#define ILQGAMES_HOST_LOGGING_HPP_

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

namespace ilqgames {
namespace host {

enum Severity { kInfo = 0, kWarning = 1, kError = 2, kFatal = 3 };

// Verbosity for VLOG(n); 0 keeps the solver quiet, as an unflagged glog run is.
inline int& VerboseLevel() {
  static int level = 0;
  return level;
}

class LogLine {
 public:
  LogLine(Severity sev, const char* file, int line) : sev_(sev) {
    static const char* tags[] = {"I", "W", "E", "F"};
    os_ << tags[sev] << " " << file << ":" << line << "] ";
  }
  ~LogLine() {
    os_ << "\n";
    std::cerr << os_.str();
    if (sev_ == kFatal) std::abort();
  }
  std::ostream& stream() { return os_; }

 private:
  Severity sev_;
  std::ostringstream os_;
};

// Swallows the stream expression of a disabled log statement.
struct LogVoidify {
  void operator&(std::ostream&) {}
};

template <typename T>
T* CheckNotNull(T* p, const char* expr, const char* file, int line) {
  if (p == nullptr) LogLine(kFatal, file, line).stream() << "Check failed: '" << expr << "' must be non NULL";
  return p;
}

}  // namespace host
}  // namespace ilqgames

#define ILQG_LOG_INFO ::ilqgames::host::kInfo
#define ILQG_LOG_WARNING ::ilqgames::host::kWarning
#define ILQG_LOG_ERROR ::ilqgames::host::kError
#define ILQG_LOG_FATAL ::ilqgames::host::kFatal

#define LOG(sev) ::ilqgames::host::LogLine(ILQG_LOG_##sev, __FILE__, __LINE__).stream()
#define LOG_IF(sev, cond) \
  !(cond) ? (void)0 : ::ilqgames::host::LogVoidify() & LOG(sev)
#define VLOG(n) LOG_IF(INFO, (n) <= ::ilqgames::host::VerboseLevel())

#define CHECK(cond) \
  (cond) ? (void)0 : ::ilqgames::host::LogVoidify() & LOG(FATAL) << "Check failed: " #cond " "
#define ILQG_CHECK_OP(a, b, op)                                                    \
  ((a)op(b)) ? (void)0                                                             \
             : ::ilqgames::host::LogVoidify() &                                    \
                   LOG(FATAL) << "Check failed: " #a " " #op " " #b " (" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) ILQG_CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) ILQG_CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) ILQG_CHECK_OP(a, b, <)
#define CHECK_LE(a, b) ILQG_CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) ILQG_CHECK_OP(a, b, >)
#define CHECK_GE(a, b) ILQG_CHECK_OP(a, b, >=)
#define CHECK_NEAR(a, b, margin) \
  CHECK(((a) > (b) ? (a) - (b) : (b) - (a)) <= (margin)) << "(" << (a) << " vs. " << (b) << ") "
#define CHECK_NOTNULL(p) ::ilqgames::host::CheckNotNull((p), #p, __FILE__, __LINE__)
#define DCHECK(cond) CHECK(cond)

// Flags: plain globals with the FLAGS_ prefix; a driver may assign them before Initialize().
#define ILQG_DEFINE_FLAG(type, name, value) type FLAGS_##name = value
#define DEFINE_double(name, value, help) ILQG_DEFINE_FLAG(double, name, value)
#define DEFINE_bool(name, value, help) ILQG_DEFINE_FLAG(bool, name, value)
#define DEFINE_int32(name, value, help) ILQG_DEFINE_FLAG(std::int32_t, name, value)
#define DEFINE_string(name, value, help) ILQG_DEFINE_FLAG(std::string, name, value)
#define DECLARE_double(name) extern double FLAGS_##name
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DECLARE_int32(name) extern std::int32_t FLAGS_##name
#define DECLARE_string(name) extern std::string FLAGS_##name

#endif  // ILQGAMES_HOST_LOGGING_HPP_
