// ComputeMatches.cpp -- drop-in for the reference's `ComputeMatches <config.yaml>` executable
// (sfm/ComputeMatches.cpp:12-68): same argv contract, YAML keys, matcher selection, stdout lines
// and SQLite side effects; the matching itself runs on an MI355X through libmsfm_match.so.
//
// Reference behaviour kept on purpose: SIFTmatch.max_distance / distance_ratio / cross_check are
// parsed but NOT passed to the matcher (sfm/ComputeMatches.cpp:38-42 vs :50,:54), so the
// constructor defaults (0.7 / 0.8 / true) are what run.  MSFM_HONOUR_YAML_MATCH_PARAMS=1 opts
// into using the YAML values instead.
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>

#include "FeatureMatching.h"
#include "Timer.h"
#include "YamlConfig.h"

using namespace MonocularSfM;

namespace {
// MSFM_CLI_TIMING=1: the wall clock (seconds since the epoch) at which main() was entered and left, on stderr -- against the caller's own
// clock around the process they bound what the loader (before) and the runtimes' exit handlers (after) take
void StampWallClock(const char* what) {
    if (!std::getenv("MSFM_CLI_TIMING")) return;
    const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::fprintf(stderr, "[msfm timing] %s at %.6f\n", what, now);
}
}  // namespace

int main(int argc, char** argv) {
    StampWallClock("main entered");
    if (argc != 2) {
        std::cout << "You need specify the YAML file path!" << std::endl;
        exit(-1);
    }
    msfm_host::YamlConfig fs;
    fs.Open(argv[1]);
    if (!fs.isOpened()) {
        std::cout << "YAML file : " << argv[1] << " can't not open!" << std::endl;
        exit(-1);
    }

    std::string database_path;
    int match_type = 1;
    double max_distance = 0.7;
    double distance_ratio = 0.8;
    bool cross_check = true;

    fs.Get("database_path", &database_path);
    fs.Get("SIFTmatch.match_type", &match_type);
    fs.Get("SIFTmatch.max_distance", &max_distance);
    fs.Get("SIFTmatch.distance_ratio", &distance_ratio);
    fs.Get("SIFTmatch.cross_check", &cross_check);

    if (!(match_type == 0 || match_type == 1)) {  // assert(match_type == 0 || match_type == 1)
        std::cerr << "ComputeMatches: SIFTmatch.match_type must be 0 (sequential) or 1 (brute)" << std::endl;
        std::abort();
    }

    const char* honour = std::getenv("MSFM_HONOUR_YAML_MATCH_PARAMS");
    const bool use_yaml = honour && honour[0] == '1';

    std::unique_ptr<FeatureMatcher> matcher;
    if (match_type == 0) {
        if (use_yaml)
            matcher.reset(new SequentialFeatureMatcher(database_path, 3, 10240, max_distance, distance_ratio, cross_check));
        else
            matcher.reset(new SequentialFeatureMatcher(database_path));
    } else {
        if (use_yaml)
            matcher.reset(new BruteFeatureMatcher(database_path, 100, true, 100, 4, 10240, max_distance, distance_ratio, cross_check));
        else
            matcher.reset(new BruteFeatureMatcher(database_path));
    }

    Timer timer;
    timer.Start();
    matcher->RunMatching();
    timer.PrintMinutes();
    matcher.reset();
    StampWallClock("main left");
    return 0;
}
