// ORACLE / TEST INFRASTRUCTURE: the few calls of CLI11 (un-vendored) that main.cpp:763-796 makes, enough to parse
//   <progMode> <inputFileName> [folderTail] [-o|--output DIR] [--log|--logLevel L] [--noProgressBar|--noPBar] [--numThreads N]
#pragma once
#include <algorithm>
#include <cctype>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
namespace CLI {
struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline std::string ignore_case(std::string s)
{
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}
struct CheckedTransformer {
    std::vector<std::pair<std::string, long long>> map;
    template <class T, class F>
    CheckedTransformer(const std::vector<std::pair<std::string, T>>& m, F)
    {
        for (const auto& e : m) map.emplace_back(ignore_case(e.first), (long long)e.second);
    }
    bool operator()(std::string& s) const
    {
        std::string k = ignore_case(s);
        for (const auto& e : map) {
            if (e.first == k || std::to_string(e.second) == k) {
                s = std::to_string(e.second);
                return true;
            }
        }
        return false;
    }
};
class Option {
public:
    std::vector<std::string> names; // "-o", "--output" or a positional name
    bool positional = false, flag = false, isRequired = false, seen = false;
    std::function<void(const std::string&)> assign;
    std::vector<CheckedTransformer> transforms;
    Option* required()
    {
        isRequired = true;
        return this;
    }
    Option* transform(const CheckedTransformer& t)
    {
        transforms.push_back(t);
        return this;
    }
    template <class T>
    Option* default_val(const T&) { return this; }
    void set(std::string v)
    {
        for (const auto& t : transforms)
            if (!t(v)) throw ParseError("invalid value '" + v + "' for " + names.front());
        assign(v);
        seen = true;
    }
};
class App {
    std::string name_;
    std::vector<std::unique_ptr<Option>> opts_;
    static std::vector<std::string> split(const std::string& s)
    {
        std::vector<std::string> r;
        std::stringstream ss(s);
        std::string t;
        while (std::getline(ss, t, ',')) r.push_back(t);
        return r;
    }
    template <class T>
    static typename std::enable_if<std::is_enum<T>::value>::type conv(const std::string& v, T& out) { out = (T)std::stoll(v); }
    template <class T>
    static typename std::enable_if<!std::is_enum<T>::value>::type conv(const std::string& v, T& out)
    {
        std::stringstream ss(v);
        ss >> out;
    }
    static void conv(const std::string& v, std::string& out) { out = v; }

public:
    App(const std::string& n)
        : name_(n) {}
    template <class T>
    Option* add_option(const std::string& names, T& var, const std::string& = "")
    {
        opts_.emplace_back(new Option);
        Option* o = opts_.back().get();
        o->names = split(names);
        o->positional = o->names.front()[0] != '-';
        o->assign = [&var](const std::string& v) { conv(v, var); };
        return o;
    }
    Option* add_flag(const std::string& names, bool& var, const std::string& = "")
    {
        opts_.emplace_back(new Option);
        Option* o = opts_.back().get();
        o->names = split(names);
        o->flag = true;
        o->assign = [&var](const std::string&) { var = true; };
        return o;
    }
    void parse(int argc, char* argv[])
    {
        size_t nextPos = 0;
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            if (!a.empty() && a[0] == '-' && a.size() > 1 && !std::isdigit((unsigned char)a[1])) {
                Option* hit = nullptr;
                for (auto& o : opts_)
                    if (!o->positional && std::find(o->names.begin(), o->names.end(), a) != o->names.end()) hit = o.get();
                if (!hit) throw ParseError("unknown option " + a);
                if (hit->flag) hit->set("1");
                else {
                    if (i + 1 >= argc) throw ParseError("missing value for " + a);
                    hit->set(argv[++i]);
                }
            }
            else {
                Option* hit = nullptr;
                size_t k = 0;
                for (auto& o : opts_)
                    if (o->positional && k++ == nextPos) hit = o.get();
                if (!hit) throw ParseError("unexpected argument " + a);
                hit->set(a);
                ++nextPos;
            }
        }
        for (auto& o : opts_)
            if (o->isRequired && !o->seen) throw ParseError("missing required " + o->names.front());
    }
    int exit(const ParseError& e) const
    {
        std::cerr << name_ << ": " << e.what() << std::endl;
        return 1;
    }
};
} // namespace CLI
