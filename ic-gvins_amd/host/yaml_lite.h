// Reader for the subset of YAML the reference's configuration uses (config/gvins.yaml: scalars, nested block maps by indentation,
// inline lists "[a, b, c]", comments) — stands in for yaml-cpp's YAML::LoadFile + node["a"]["b"].as<T>() (ic_gvins.cc:51-144,
// ROS/fusion_ros.cc:63-96).  Keys of nested maps are joined with '.', e.g. "cam0.intrinsic", "imumodel.arw".
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace icg {

class YamlLite {
public:
    static bool load(const std::string &path, YamlLite &out, std::string *err = nullptr) {
        std::ifstream f(path);
        if (!f) {
            if (err) *err = "cannot open " + path;
            return false;
        }
        out.values_.clear();
        out.text_.clear();
        std::vector<std::pair<int, std::string>> stack; // (indent, key prefix)
        std::string line;
        while (std::getline(f, line)) {
            out.text_ += line + "\n";
            bool quoted = false;
            for (size_t i = 0; i < line.size(); i++) {
                if (line[i] == '"' || line[i] == '\'') quoted = !quoted;
                if (line[i] == '#' && !quoted) {
                    line = line.substr(0, i);
                    break;
                }
            }
            size_t indent = line.find_first_not_of(" \t");
            if (indent == std::string::npos) continue;
            size_t colon = line.find(':', indent);
            if (colon == std::string::npos) continue;
            std::string key = trim(line.substr(indent, colon - indent)), val = trim(line.substr(colon + 1));
            while (!stack.empty() && stack.back().first >= (int) indent) stack.pop_back();
            std::string full = stack.empty() ? key : stack.back().second + "." + key;
            if (val.empty())
                stack.emplace_back((int) indent, full);
            else
                out.values_[full] = unquote(val);
        }
        return true;
    }
    bool has(const std::string &key) const { return values_.count(key) != 0; }
    // node[key].as<T>(): a missing key throws like yaml-cpp's BadConversion on a null node
    std::string str(const std::string &key) const {
        auto it = values_.find(key);
        if (it == values_.end()) throw std::runtime_error("configuration key missing: " + key);
        return it->second;
    }
    double real(const std::string &key) const { return number(str(key), key); }
    long integer(const std::string &key) const { return (long) number(str(key), key); }
    bool boolean(const std::string &key) const {
        std::string v = str(key);
        return v == "true" || v == "True" || v == "TRUE" || v == "1" || v == "yes";
    }
    std::vector<double> reals(const std::string &key) const {
        std::string v = str(key);
        std::vector<double> out;
        size_t a = v.find('['), b = v.rfind(']');
        if (a == std::string::npos || b == std::string::npos || b < a) throw std::runtime_error("configuration key is not a list: " + key);
        std::string body = v.substr(a + 1, b - a - 1);
        size_t pos       = 0;
        while (pos < body.size()) {
            size_t comma    = body.find(',', pos);
            std::string tok = trim(body.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos));
            if (!tok.empty()) out.push_back(number(tok, key));
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        return out;
    }
    const std::string &text() const { return text_; } // the file as read (the reference dumps a copy into the output directory)

private:
    // yaml-cpp throws BadConversion on a scalar that is not a number; a silent 0 here once hid a malformed extrinsic
    static double number(const std::string &tok, const std::string &key) {
        char *end = nullptr;
        double v  = strtod(tok.c_str(), &end);
        if (end == tok.c_str() || (end && *end != 0)) throw std::runtime_error("configuration key " + key + ": '" + tok + "' is not a number");
        return v;
    }
    static std::string trim(const std::string &s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static std::string unquote(const std::string &s) {
        if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
        return s;
    }
    std::map<std::string, std::string> values_;
    std::string text_;
};

} // namespace icg
