// ORACLE / TEST INFRASTRUCTURE: logging calls of the reference compile to nothing.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
// {fmt} comes with spdlog; the reference formats a few file names with it.  Enough of the grammar for those:
// "{}", "{:d}", "{:g}", "{:s}", "{:05d}" -- each replacement field consumes one argument.
namespace fmt {
namespace detail {
inline void put(std::ostringstream& os, const std::string& spec, const std::string& v) { (void)spec; os << v; }
inline void put(std::ostringstream& os, const std::string& spec, const char* v) { (void)spec; os << v; }
template <class T>
inline void put(std::ostringstream& os, const std::string& spec, const T& v)
{
    if (spec.empty()) { os << v; return; }
    char buf[128];
    std::string f = "%" + spec;
    char c = spec.back();
    if (c == 'd') { f = "%" + spec.substr(0, spec.size() - 1) + "lld"; std::snprintf(buf, sizeof buf, f.c_str(), (long long)v); }
    else if (c == 'g' || c == 'f' || c == 'e') std::snprintf(buf, sizeof buf, f.c_str(), (double)v);
    else { os << v; return; }
    os << buf;
}
inline void run(std::ostringstream& os, const char* f) { os << f; }
template <class T, class... R>
inline void run(std::ostringstream& os, const char* f, const T& v, const R&... rest)
{
    for (; *f; ++f) {
        if (*f == '{' && f[1] == '{') { os << '{'; ++f; continue; }
        if (*f == '}' && f[1] == '}') { os << '}'; ++f; continue; }
        if (*f == '{') {
            const char* e = f;
            while (*e && *e != '}') ++e;
            std::string spec(f + 1, e);
            if (!spec.empty() && spec[0] == ':') spec = spec.substr(1);
            put(os, spec, v);
            run(os, *e ? e + 1 : e, rest...);
            return;
        }
        os << *f;
    }
}
} // namespace detail
template <class... A>
inline std::string format(const char* f, const A&... a)
{
    std::ostringstream os;
    detail::run(os, f, a...);
    return os.str();
}
template <class... A>
inline std::string format(const std::string& f, const A&... a) { return format(f.c_str(), a...); }
// the progress bar of main.cpp (positional fields): not rendered
template <class... A>
inline void print(const char*, const A&...) {}
} // namespace fmt
namespace spdlog {
namespace level { enum level_enum { trace, debug, info, warn, err, critical, off }; }
// messages are dropped unless IPCREF_LOG is set in the environment (debugging the stand-ins)
inline bool refshim_log_on()
{
    static const bool on = std::getenv("IPCREF_LOG") != nullptr;
    return on;
}
template <class... A>
inline void refshim_log(const char* lvl, const char* f, const A&... a)
{
    if (refshim_log_on()) std::fprintf(stderr, "[%s] %s\n", lvl, fmt::format(f, a...).c_str());
}
template <class... A> inline void trace(const char* f, const A&... a) { refshim_log("trace", f, a...); }
template <class... A> inline void debug(const char* f, const A&... a) { refshim_log("debug", f, a...); }
template <class... A> inline void info(const char* f, const A&... a) { refshim_log("info", f, a...); }
template <class... A> inline void warn(const char* f, const A&... a) { refshim_log("warn", f, a...); }
template <class... A> inline void error(const char* f, const A&... a) { refshim_log("error", f, a...); }
template <class... A> inline void critical(const char* f, const A&... a) { refshim_log("critical", f, a...); }
template <class... A> inline void info(const std::string& f, const A&... a) { refshim_log("info", f.c_str(), a...); }
template <class... A> inline void error(const std::string& f, const A&... a) { refshim_log("error", f.c_str(), a...); }
inline void set_level(level::level_enum) {}
inline level::level_enum get_level() { return level::off; }
} // namespace spdlog
