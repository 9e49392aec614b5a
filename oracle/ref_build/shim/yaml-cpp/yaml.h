// ORACLE / TEST INFRASTRUCTURE ONLY — flat "key: value" reader with the YAML::LoadFile / node["key"].as<T>() surface the
// reference tracker constructor uses (tracking.cc:49-61)
#pragma once
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
namespace YAML {
class Node {
public:
    Node() {}
    explicit Node(std::string scalar) : scalar_(std::move(scalar)) {}
    Node operator[](const std::string &key) const {
        auto it = map_.find(key);
        if (it == map_.end()) throw std::runtime_error("yaml shim: missing key " + key);
        return Node(it->second);
    }
    template <typename T> T as() const;
    std::map<std::string, std::string> map_;
    std::string scalar_;
};
template <> inline bool Node::as<bool>() const { return scalar_ == "true" || scalar_ == "True" || scalar_ == "1"; }
template <> inline int Node::as<int>() const { return std::stoi(scalar_); }
template <> inline double Node::as<double>() const { return std::stod(scalar_); }
template <> inline std::string Node::as<std::string>() const { return scalar_; }
inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n\"'"), b = s.find_last_not_of(" \t\r\n\"'");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
inline Node LoadFile(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("yaml shim: cannot open " + path);
    Node n;
    std::string line;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        size_t c = line.find(':');
        if (c == std::string::npos) continue;
        n.map_[trim(line.substr(0, c))] = trim(line.substr(c + 1));
    }
    return n;
}
} // namespace YAML
