#include "YamlConfig.h"

#include <cstdlib>
#include <fstream>

namespace msfm_host {

namespace {
std::string Trim(const std::string& s) {
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
}
}  // namespace

bool YamlConfig::Open(const std::string& path) {
    opened_ = false;
    values_.clear();
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    bool first = true;
    while (std::getline(in, line)) {
        if (first) {
            first = false;
            // cv::FileStorage requires the YAML directive on the first line
            if (Trim(line).rfind("%YAML", 0) != 0) return false;
            continue;
        }
        // strip comments that are not inside a quoted string
        bool in_quote = false;
        std::string body;
        for (char c : line) {
            if (c == '"') in_quote = !in_quote;
            if (c == '#' && !in_quote) break;
            body.push_back(c);
        }
        body = Trim(body);
        if (body.empty() || body == "---" || body == "...") continue;
        const size_t colon = body.find(':');
        if (colon == std::string::npos) continue;
        std::string key = Trim(body.substr(0, colon));
        std::string val = Trim(body.substr(colon + 1));
        if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
        if (!key.empty()) values_[key] = val;
    }
    opened_ = true;
    return true;
}

void YamlConfig::Get(const std::string& key, std::string* out) const {
    auto it = values_.find(key);
    *out = (it == values_.end()) ? std::string() : it->second;
}

void YamlConfig::Get(const std::string& key, int* out) const {
    auto it = values_.find(key);
    if (it == values_.end() || it->second.empty()) return;
    *out = (int)std::strtod(it->second.c_str(), nullptr);
}

void YamlConfig::Get(const std::string& key, double* out) const {
    auto it = values_.find(key);
    if (it == values_.end() || it->second.empty()) return;
    *out = std::strtod(it->second.c_str(), nullptr);
}

void YamlConfig::Get(const std::string& key, bool* out) const {
    int v = *out ? 1 : 0;
    Get(key, &v);
    *out = (v != 0);
}

}  // namespace msfm_host
