// TEST-ONLY stand-in for lms/module.h (see ../README.md).  Surface inferred from the reference's
// usage: include/trajectory_point_follower.h:26-30,45,59-68; src/trajectory_point_follower.cpp:9-13,
// :25, :35, :64, :80, :134, :292.
#pragma once
#include <chrono>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <typeindex>
#include <vector>

namespace lms {

class Time {
public:
    static const Time ZERO;
    static Time now() { return Time(std::chrono::duration_cast<std::chrono::microseconds>(
                                         std::chrono::steady_clock::now().time_since_epoch()).count()); }
    static Time fromMillis(long long ms) { return Time(ms * 1000); }
    Time since() const { return Time(now().us_ - us_); }
    bool operator>(const Time& o) const { return us_ > o.us_; }
    explicit Time(long long us = 0) : us_(us) {}
private:
    long long us_;
};
inline const Time Time::ZERO = Time(0);

// config().get<T>(key, default) / getArray<T>(key)
class Config {
public:
    template <typename T> T get(const std::string& key, const T& def = T()) const {
        auto it = values_.find(key);
        if (it == values_.end()) return def;
        std::istringstream is(it->second);
        T out = def;
        if constexpr (std::is_same<T, std::string>::value) return it->second;
        else { is >> out; return out; }
    }
    template <typename T> std::vector<T> getArray(const std::string& key) const {
        std::vector<T> out;
        auto it = values_.find(key);
        if (it == values_.end()) return out;
        std::istringstream is(it->second);
        std::string tok;
        while (std::getline(is, tok, ',')) { std::istringstream ts(tok); T x; if (ts >> x) out.push_back(x); }
        return out;
    }
    template <typename T> void set(const std::string& key, const T& val) {
        std::ostringstream os; os.precision(17); os << val; values_[key] = os.str();
    }
private:
    std::map<std::string, std::string> values_;
};

// logger.debug("tag") << ... ; also warn/error/time/timeEnd
class LogLine {
public:
    explicit LogLine(std::ostream* os) : os_(os) {}
    template <typename T> LogLine& operator<<(const T& x) { if (os_) (*os_) << x; return *this; }
    ~LogLine() { if (os_) (*os_) << '\n'; }
private:
    std::ostream* os_;
};
class Logger {
public:
    bool verbose = false;
    LogLine debug(const std::string& tag = "") { return line(verbose, "DEBUG", tag); }
    LogLine warn(const std::string& tag = "") { return line(verbose, "WARN", tag); }
    LogLine error(const std::string& tag = "") { return line(true, "ERROR", tag); }
    void time(const std::string&) {}
    void timeEnd(const std::string&) {}
private:
    LogLine line(bool on, const char* lvl, const std::string& tag) {
        if (!on) return LogLine(nullptr);
        std::cerr << lvl << ' ' << tag << ": ";
        return LogLine(&std::cerr);
    }
};

// Data channels: named, typed, shared between modules of one runtime.
class ChannelStore {
public:
    template <typename T> std::shared_ptr<T> get(const std::string& name) {
        auto& slot = slots_[name];
        if (!slot) slot = std::static_pointer_cast<void>(std::make_shared<T>());
        return std::static_pointer_cast<T>(slot);
    }
private:
    std::map<std::string, std::shared_ptr<void>> slots_;
};
template <typename T> class ReadDataChannel {
public:
    ReadDataChannel() = default;
    explicit ReadDataChannel(std::shared_ptr<T> p) : p_(std::move(p)) {}
    const T* operator->() const { return p_.get(); }
    const T& operator*() const { return *p_; }
private:
    std::shared_ptr<T> p_;
};
template <typename T> class WriteDataChannel {
public:
    WriteDataChannel() = default;
    explicit WriteDataChannel(std::shared_ptr<T> p) : p_(std::move(p)) {}
    T* operator->() const { return p_.get(); }
    T& operator*() const { return *p_; }
private:
    std::shared_ptr<T> p_;
};

class Module {
public:
    virtual ~Module() = default;
    virtual bool initialize() = 0;
    virtual bool deinitialize() = 0;
    virtual bool cycle() = 0;
    virtual void configsChanged() {}

    // test harness wiring (the real runtime does this when it loads the module)
    void attach(ChannelStore* channels, std::map<std::string, std::shared_ptr<void>>* services) {
        channels_ = channels; services_ = services;
    }
    Config& config() { return config_; }
    Logger logger;

protected:
    template <typename T> ReadDataChannel<T> readChannel(const std::string& name) {
        return ReadDataChannel<T>(channels_->get<T>(name));
    }
    template <typename T> WriteDataChannel<T> writeChannel(const std::string& name) {
        return WriteDataChannel<T>(channels_->get<T>(name));
    }
    template <typename T> std::shared_ptr<T> getService(const std::string& name) {
        return std::static_pointer_cast<T>((*services_)[name]);
    }

private:
    Config config_;
    ChannelStore* channels_ = nullptr;
    std::map<std::string, std::shared_ptr<void>>* services_ = nullptr;
};

}  // namespace lms

// reference: src/interface.cpp:3
#define LMS_MODULE_INTERFACE(CLASS) \
    extern "C" { void* getInstance() { return new CLASS(); } }
