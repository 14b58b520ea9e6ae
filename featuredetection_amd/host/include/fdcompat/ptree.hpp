// fdcompat/ptree.hpp -- the subset of boost::property_tree (INFO format) that the reference's
// config loading uses (ffpDetectApp.cpp:375-500, *.cfg files): nested "key value { children }"
// nodes, ';' comments, quoted strings, get<T>(path[, default]), get_child, iteration.
// Define FD_USE_BOOST to use the real library instead.
#pragma once
#ifdef FD_USE_BOOST
#include "boost/property_tree/ptree.hpp"
#include "boost/property_tree/info_parser.hpp"
#else
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace boost { namespace property_tree {

struct ptree_error : std::runtime_error { using std::runtime_error::runtime_error; };

class ptree {
public:
    typedef std::pair<std::string, ptree> value_type;
    typedef std::vector<value_type>::const_iterator const_iterator;
    std::string data;
    std::vector<value_type> children;

    const_iterator begin() const { return children.begin(); }
    const_iterator end() const { return children.end(); }
    bool empty() const { return children.empty(); }
    size_t count(const std::string& key) const {
        size_t n = 0;
        for (const auto& c : children) n += c.first == key;
        return n;
    }

    const ptree* find(const std::string& path) const {
        const ptree* cur = this;
        size_t pos = 0;
        while (pos <= path.size()) {
            size_t dot = path.find('.', pos);
            std::string key = path.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos);
            const ptree* next = nullptr;
            for (const auto& c : cur->children) if (c.first == key) { next = &c.second; break; }
            if (!next) return nullptr;
            cur = next;
            if (dot == std::string::npos) break;
            pos = dot + 1;
        }
        return cur;
    }
    const ptree& get_child(const std::string& path) const {
        const ptree* p = find(path);
        if (!p) throw ptree_error("No such node (" + path + ")");
        return *p;
    }
    template <class T> T get_value() const {
        std::istringstream ss(data);
        T v;
        if (!(ss >> v)) throw ptree_error("conversion of data \"" + data + "\" failed");
        return v;
    }
    template <class T> T get(const std::string& path) const { return get_child(path).get_value<T>(); }
    template <class T> T get(const std::string& path, const T& def) const {
        const ptree* p = find(path);
        if (!p) return def;
        try { return p->get_value<T>(); } catch (const ptree_error&) { return def; }
    }
    void put(const std::string& key, const std::string& value) { ptree c; c.data = value; children.emplace_back(key, c); }
};
template <> inline std::string ptree::get_value<std::string>() const { return data; }

namespace detail {
inline bool next_token(std::istream& in, std::string& tok, bool& eol) {
    tok.clear();
    eol = false;
    int c;
    while ((c = in.peek()) != EOF) {
        if (c == '\n') { in.get(); eol = true; return false; }
        if (c == ' ' || c == '\t' || c == '\r') { in.get(); continue; }
        if (c == ';') { while ((c = in.get()) != EOF && c != '\n') {} eol = true; return false; }
        break;
    }
    if (c == EOF) { eol = true; return false; }
    if (c == '"') {
        in.get();
        while ((c = in.get()) != EOF && c != '"') { if (c == '\\') c = in.get(); tok += (char)c; }
        return true;
    }
    if (c == '{' || c == '}') { tok = (char)in.get(); return true; }
    while ((c = in.peek()) != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r' && c != ';' && c != '{' && c != '}') tok += (char)in.get();
    return true;
}
inline void parse_block(std::istream& in, ptree& node, bool top) {
    std::string tok;
    bool eol;
    ptree* last = nullptr;
    while (in.peek() != EOF) {
        if (!next_token(in, tok, eol)) continue;
        if (tok == "}") { if (top) throw ptree_error("unmatched '}'"); return; }
        if (tok == "{") { if (!last) throw ptree_error("unexpected '{'"); parse_block(in, *last, false); continue; }
        node.children.emplace_back(tok, ptree());
        last = &node.children.back().second;
        std::string val;
        if (next_token(in, val, eol)) {
            if (val == "{") parse_block(in, *last, false);
            else if (val == "}") { if (top) throw ptree_error("unmatched '}'"); return; }
            else last->data = val;
        }
    }
    if (!top) throw ptree_error("missing '}'");
}
}  // namespace detail

inline void read_info(std::istream& in, ptree& pt) { pt = ptree(); detail::parse_block(in, pt, true); }
inline void read_info(const std::string& filename, ptree& pt) {
    std::ifstream f(filename.c_str());
    if (!f.is_open()) throw ptree_error("cannot open " + filename);
    read_info(f, pt);
}

}}  // namespace boost::property_tree
#endif
