// Timer.h -- stopwatch with the reference's output format (src/Common/Timer.cpp:58-109):
// "Elapsed time: %.5f [seconds|minutes|hours]" followed by a newline.
#pragma once
#include <chrono>
#include <iomanip>
#include <cstdio>
#include <iostream>
#include <string>

namespace MonocularSfM {

class Timer {
public:
    void Start() {
        started_ = true;
        paused_ = false;
        start_time_ = std::chrono::high_resolution_clock::now();
    }
    void Restart() { Start(); }
    void Pause() {
        paused_ = true;
        pause_time_ = std::chrono::high_resolution_clock::now();
    }
    void Resume() {
        paused_ = false;
        start_time_ += std::chrono::high_resolution_clock::now() - pause_time_;
    }
    double ElapsedMicroSeconds() const {
        if (!started_) return 0.0;
        const auto end = paused_ ? pause_time_ : std::chrono::high_resolution_clock::now();
        return (double)std::chrono::duration_cast<std::chrono::microseconds>(end - start_time_).count();
    }
    double ElapsedSeconds() const { return ElapsedMicroSeconds() / 1e6; }
    double ElapsedMinutes() const { return ElapsedSeconds() / 60; }
    double ElapsedHours() const { return ElapsedMinutes() / 60; }
    void PrintSeconds() const { Print(ElapsedSeconds(), "seconds"); }
    void PrintMinutes() const { Print(ElapsedMinutes(), "minutes"); }
    void PrintHours() const { Print(ElapsedHours(), "hours"); }
    static void Print(double v, const char* unit) { std::cout << Format(v, unit) << std::flush; }
    // the line Print writes, newline included
    static std::string Format(double v, const char* unit) {
        char buf[96];
        std::snprintf(buf, sizeof(buf), "Elapsed time: %.5f [%s]\n", v, unit);
        return buf;
    }

private:
    bool started_ = false, paused_ = false;
    std::chrono::high_resolution_clock::time_point start_time_, pause_time_;
};

}  // namespace MonocularSfM
