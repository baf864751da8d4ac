// ros_params.hpp — the parameter interface the reference's nodes use, without ROS2.
//
// The reference reads every setting through rclcpp's declare_parameter / get_parameter(...).as_*()
// (src/decentral_legged_est/src/EstSub.cpp:123-208, src/orien_est/src/orien_ekf.cpp:13-25) from a
// ROS2 parameter file (src/go1_example/config/parameters_go1.yaml: `<node>: ros__parameters: ...`).
// ParamNode offers the same two calls over the same file format, so the parameter wrappers in
// est_node_core.hpp / orien_node_core.hpp are written once as templates over "something with
// declare_parameter/get_parameter": a ParamNode here, the rclcpp::Node itself in a ROS2 build.
//
// File format understood (the subset ROS2 parameter files use): nested block mappings by indentation,
// `#` comments, scalars (bool, int, double, quoted or bare string) and flow sequences `[a, b, c]`
// that may continue over several lines and may start on the line after their key.
// Nested keys become dotted names (`prior.p_init_std`).  A file value overrides the declared default;
// integers are accepted where a double (array) was declared (rclcpp would reject them).
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace dekf_ros {

class ParamValue {
  public:
    enum Type { NOT_SET, BOOL, INT, DOUBLE, STRING, DOUBLE_ARRAY, STRING_ARRAY };
    ParamValue() {}
    static ParamValue of(bool v) { ParamValue p; p.t_ = BOOL; p.i_ = v; return p; }
    static ParamValue of(long v) { ParamValue p; p.t_ = INT; p.i_ = v; return p; }
    static ParamValue of(double v) { ParamValue p; p.t_ = DOUBLE; p.d_ = v; return p; }
    static ParamValue of(const std::string& v) { ParamValue p; p.t_ = STRING; p.s_ = v; return p; }
    static ParamValue of(const std::vector<double>& v, bool all_int = false) {
        ParamValue p; p.t_ = DOUBLE_ARRAY; p.a_ = v; p.all_int_ = all_int; return p;
    }
    static ParamValue of(const std::vector<std::string>& v) { ParamValue p; p.t_ = STRING_ARRAY; p.sa_ = v; return p; }

    Type type() const { return t_; }
    bool as_bool() const { need(BOOL, "bool"); return i_ != 0; }
    long as_int() const { need(INT, "int"); return i_; }
    double as_double() const { need(DOUBLE, "double"); return d_; }
    const std::string& as_string() const { need(STRING, "string"); return s_; }
    const std::vector<double>& as_double_array() const { need(DOUBLE_ARRAY, "double array"); return a_; }
    const std::vector<std::string>& as_string_array() const { need(STRING_ARRAY, "string array"); return sa_; }

    // value of the file coerced to the type of the declared default
    ParamValue coerced_to(Type want, const std::string& name) const {
        if (t_ == want) return *this;
        if (t_ == INT && want == DOUBLE) return of((double)i_);
        if (t_ == DOUBLE_ARRAY && want == DOUBLE_ARRAY) return *this;
        throw std::invalid_argument("parameter '" + name + "': the file's value has another type than the declared default");
    }

  private:
    void need(Type t, const char* what) const {
        if (t_ != t) throw std::invalid_argument(std::string("parameter is not a ") + what);
    }
    Type t_ = NOT_SET;
    long i_ = 0;
    double d_ = 0.0;
    std::string s_;
    std::vector<double> a_;
    std::vector<std::string> sa_;
    bool all_int_ = false;
};

namespace detail {
inline std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
// cut a trailing comment: '#' at the start or after white space, outside quotes
inline std::string strip_comment(const std::string& s) {
    char quote = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (quote) { if (c == quote) quote = 0; continue; }
        if (c == '"' || c == '\'') quote = c;
        else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}
inline int bracket_balance(const std::string& s) {
    int b = 0;
    char quote = 0;
    for (char c : s) {
        if (quote) { if (c == quote) quote = 0; continue; }
        if (c == '"' || c == '\'') quote = c;
        else if (c == '[') ++b;
        else if (c == ']') --b;
    }
    return b;
}
inline bool parse_long(const std::string& s, long& v) {
    if (s.empty()) return false;
    char* end = nullptr;
    v = std::strtol(s.c_str(), &end, 10);
    return end && *end == 0 && s.find_first_of("0123456789") != std::string::npos;
}
inline bool parse_double(const std::string& s, double& v) {
    if (s.empty() || s.find_first_of("0123456789") == std::string::npos) return false;
    char* end = nullptr;
    v = std::strtod(s.c_str(), &end);
    return end && *end == 0;
}
inline ParamValue parse_scalar(const std::string& raw) {
    const std::string s = trim(raw);
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        return ParamValue::of(s.substr(1, s.size() - 2));
    if (s == "true" || s == "True" || s == "TRUE") return ParamValue::of(true);
    if (s == "false" || s == "False" || s == "FALSE") return ParamValue::of(false);
    long l;
    if (parse_long(s, l)) return ParamValue::of(l);
    double d;
    if (parse_double(s, d)) return ParamValue::of(d);
    return ParamValue::of(s);
}
inline ParamValue parse_flow_sequence(const std::string& raw, const std::string& name) {
    const std::string s = trim(raw);
    if (s.size() < 2 || s.front() != '[' || s.back() != ']') throw std::invalid_argument("parameter '" + name + "': malformed sequence");
    std::vector<std::string> items;
    std::string cur;
    char quote = 0;
    for (size_t i = 1; i + 1 < s.size(); ++i) {
        const char c = s[i];
        if (quote) { cur += c; if (c == quote) quote = 0; continue; }
        if (c == '"' || c == '\'') { quote = c; cur += c; }
        else if (c == ',') { items.push_back(trim(cur)); cur.clear(); }
        else cur += c;
    }
    if (!trim(cur).empty()) items.push_back(trim(cur));
    std::vector<double> nums;
    bool numeric = !items.empty(), all_int = true;
    for (const std::string& it : items) {
        long l;
        double d;
        if (parse_long(it, l)) nums.push_back((double)l);
        else if (parse_double(it, d)) { nums.push_back(d); all_int = false; }
        else { numeric = false; break; }
    }
    if (numeric) return ParamValue::of(nums, all_int);
    std::vector<std::string> strs;
    for (const std::string& it : items) {
        ParamValue v = parse_scalar(it);
        strs.push_back(v.type() == ParamValue::STRING ? v.as_string() : it);
    }
    return ParamValue::of(strs);
}
}  // namespace detail

// Every parameter of one file as `node.ros__parameters.a.b` -> value
inline std::map<std::string, ParamValue> parse_parameter_text(const std::string& text) {
    using namespace detail;
    std::map<std::string, ParamValue> out;
    std::vector<std::pair<int, std::string>> stack;  // open mappings: (indent of their key, dotted name)
    std::istringstream in(text);
    std::string line;
    int lineno = 0;
    auto fail = [&](const std::string& why) { throw std::invalid_argument("parameter file line " + std::to_string(lineno) + ": " + why); };
    auto next_content = [&](std::string& dst) {
        while (std::getline(in, line)) {
            ++lineno;
            dst = strip_comment(line);
            if (!trim(dst).empty()) return true;
        }
        return false;
    };
    std::string cur;
    while (next_content(cur)) {
        if (cur.find('\t') != std::string::npos && cur.find_first_not_of(" \t") > cur.find('\t')) fail("tab in the indentation");
        const int indent = (int)cur.find_first_not_of(' ');
        std::string body = trim(cur);
        if (body.front() == '[') {
            // a flow sequence that starts on the line after its key
            if (stack.empty()) fail("sequence without a key");
            const std::string name = stack.back().second;
            stack.pop_back();
            std::string more;
            while (bracket_balance(body) > 0) { if (!next_content(more)) fail("unterminated sequence"); body += " " + trim(more); }
            out[name] = parse_flow_sequence(body, name);
            continue;
        }
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        size_t colon = std::string::npos;
        for (size_t i = 0; i < body.size(); ++i)
            if (body[i] == ':' && (i + 1 == body.size() || body[i + 1] == ' ')) { colon = i; break; }
        if (colon == std::string::npos) fail("expected 'key: value'");
        std::string key = trim(body.substr(0, colon));
        if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
        std::string value = trim(body.substr(colon + 1));
        const std::string name = stack.empty() ? key : stack.back().second + "." + key;
        if (value.empty()) { stack.emplace_back(indent, name); continue; }
        if (value == "{}") continue;  // empty mapping
        if (value.front() == '[') {
            std::string more;
            while (bracket_balance(value) > 0) { if (!next_content(more)) fail("unterminated sequence"); value += " " + trim(more); }
            out[name] = parse_flow_sequence(value, name);
        } else {
            out[name] = parse_scalar(value);
        }
    }
    return out;
}

// One node's view of a parameter file, with rclcpp::Node's two calls.
class ParamNode {
  public:
    ParamNode() {}
    // parameters under `<node_name>: ros__parameters:` (a leading '/' in the file is ignored) and under
    // the wildcard `/**: ros__parameters:`; the named node wins
    static ParamNode from_text(const std::string& text, const std::string& node_name) {
        ParamNode n;
        const auto all = parse_parameter_text(text);
        for (const char* who : {"/**", node_name.c_str()}) {
            for (const std::string lead : {"", "/"}) {
                const std::string prefix = lead + who + ".ros__parameters.";
                for (const auto& kv : all)
                    if (kv.first.compare(0, prefix.size(), prefix) == 0) n.file_[kv.first.substr(prefix.size())] = kv.second;
            }
        }
        return n;
    }
    static ParamNode from_file(const std::string& path, const std::string& node_name) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open parameter file " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        return from_text(ss.str(), node_name);
    }

    void declare_parameter(const std::string& name, bool def) { declare(name, ParamValue::of(def)); }
    void declare_parameter(const std::string& name, int def) { declare(name, ParamValue::of((long)def)); }
    void declare_parameter(const std::string& name, long def) { declare(name, ParamValue::of(def)); }
    void declare_parameter(const std::string& name, double def) { declare(name, ParamValue::of(def)); }
    void declare_parameter(const std::string& name, const char* def) { declare(name, ParamValue::of(std::string(def))); }
    void declare_parameter(const std::string& name, const std::string& def) { declare(name, ParamValue::of(def)); }
    void declare_parameter(const std::string& name, const std::vector<double>& def) { declare(name, ParamValue::of(def)); }

    const ParamValue& get_parameter(const std::string& name) const {
        auto it = declared_.find(name);
        if (it == declared_.end()) throw std::invalid_argument("parameter '" + name + "' was not declared");
        return it->second;
    }
    bool has_override(const std::string& name) const { return file_.count(name) != 0; }
    // names the file sets but nobody declared (rclcpp ignores those silently; useful for a typo check)
    std::vector<std::string> undeclared_overrides() const {
        std::vector<std::string> r;
        for (const auto& kv : file_) if (!declared_.count(kv.first)) r.push_back(kv.first);
        return r;
    }

  private:
    void declare(const std::string& name, const ParamValue& def) {
        auto it = file_.find(name);
        declared_[name] = it == file_.end() ? def : it->second.coerced_to(def.type(), name);
    }
    std::map<std::string, ParamValue> file_, declared_;
};

}  // namespace dekf_ros
