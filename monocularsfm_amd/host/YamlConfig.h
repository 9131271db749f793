// YamlConfig.h -- reader for the subset of OpenCV FileStorage YAML 1.0 the reference's configs use
// (config/*.yaml: a "%YAML:1.0" first line, flat "key : value" scalars whose keys may contain dots,
// '#' comments, double-quoted strings).  Replaces cv::FileStorage in sfm/ComputeMatches.cpp:24-42.
#pragma once
#include <map>
#include <string>

namespace msfm_host {

class YamlConfig {
public:
    // false if the file cannot be opened or does not start with the %YAML directive
    // (cv::FileStorage::isOpened() is false for both).
    bool Open(const std::string& path);
    bool isOpened() const { return opened_; }
    bool Has(const std::string& key) const { return values_.count(key) != 0; }
    // Like `fs["key"] >> var`: a missing key leaves *out untouched... except for strings, which
    // cv::FileNode >> std::string resets to empty.
    void Get(const std::string& key, std::string* out) const;
    void Get(const std::string& key, int* out) const;
    void Get(const std::string& key, double* out) const;
    void Get(const std::string& key, bool* out) const;

private:
    bool opened_ = false;
    std::map<std::string, std::string> values_;
};

}  // namespace msfm_host
