// ORACLE / TEST INFRASTRUCTURE ONLY — the yaml-cpp surface the reference uses (YAML::LoadFile, node["a"]["b"].as<T>(), YAML::Dump,
// YAML::Exception; tracking.cc:49-61, ic_gvins.cc:51-144) over the subset of YAML its configuration needs: scalars, block maps nested by
// indentation, inline lists "[a, b, c]", comments.
#pragma once
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace YAML {
class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string &what) : std::runtime_error(what) {}
};
inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n\"'"), b = s.find_last_not_of(" \t\r\n\"'");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
class Node {
public:
    Node() : d_(std::make_shared<Data>()) {}
    Node operator[](const std::string &key) const {
        auto it = d_->map.find(key);
        if (it == d_->map.end()) throw Exception("yaml shim: missing key " + key);
        return it->second;
    }
    template <typename T> T as() const { return convert((T *) nullptr); }
    struct Data {
        std::map<std::string, Node> map;
        std::string scalar, text;
    };
    std::shared_ptr<Data> d_;

private:
    bool convert(bool *) const { return d_->scalar == "true" || d_->scalar == "True" || d_->scalar == "1"; }
    int convert(int *) const { return std::stoi(d_->scalar); }
    long convert(long *) const { return std::stol(d_->scalar); }
    unsigned long convert(unsigned long *) const { return std::stoul(d_->scalar); }
    double convert(double *) const { return std::stod(d_->scalar); }
    std::string convert(std::string *) const { return d_->scalar; }
    template <typename E> std::vector<E> convert(std::vector<E> *) const {
        std::vector<E> out;
        size_t a = d_->scalar.find('['), b = d_->scalar.rfind(']');
        if (a == std::string::npos || b == std::string::npos) throw Exception("yaml shim: not a list: " + d_->scalar);
        std::stringstream ss(d_->scalar.substr(a + 1, b - a - 1));
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            tok = trim(tok);
            if (!tok.empty()) out.push_back((E) std::stod(tok));
        }
        return out;
    }
};
inline Node LoadFile(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw Exception("yaml shim: cannot open " + path);
    Node root;
    std::vector<std::pair<int, Node>> stack; // (indent, map node)
    std::string line;
    while (std::getline(f, line)) {
        root.d_->text += line + "\n";
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        size_t indent = line.find_first_not_of(" \t");
        if (indent == std::string::npos) continue;
        size_t c = line.find(':', indent);
        if (c == std::string::npos) continue;
        std::string key = trim(line.substr(indent, c - indent)), val = trim(line.substr(c + 1));
        while (!stack.empty() && stack.back().first >= (int) indent) stack.pop_back();
        Node parent = stack.empty() ? root : stack.back().second;
        Node child;
        child.d_->scalar    = val;
        parent.d_->map[key] = child;
        if (val.empty()) stack.emplace_back((int) indent, child);
    }
    return root;
}
inline std::string Dump(const Node &n) { return n.d_->text; }
} // namespace YAML
