// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN for the boost.thread / boost.bind / boost.function vocabulary the
// reference's hot-path sources use (util/IndexThreadReduce.h, DataStructures/Frame*.h): mapped onto the C++17 standard
// library, which provides the same primitives.  boost is an external dependency absent from this machine.
#pragma once
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <sys/time.h>   // the platform headers of boost.thread bring gettimeofday along

namespace boost {
using std::thread;
using std::mutex;
using std::recursive_mutex;
using std::condition_variable;
using std::unique_lock;
using std::shared_lock;
using std::lock_guard;
using std::function;
using std::bind;
namespace posix_time {
inline std::chrono::milliseconds milliseconds(long n) { return std::chrono::milliseconds(n); }
}
// boost::shared_mutex::timed_lock(duration) -> std::shared_timed_mutex::try_lock_for
class shared_mutex : public std::shared_timed_mutex {
 public:
  template <typename D> bool timed_lock(const D& d) { return try_lock_for(d); }
};
}  // namespace boost
using namespace std::placeholders;   // boost/bind.hpp puts _1, _2, ... in the global namespace
