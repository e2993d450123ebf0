// cs_config.h: the CS_* switches, read from the environment once.
#include "cs_config.h"

#include <cstdlib>

#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <string_view>

extern char** environ;

namespace cs {
namespace {
struct Store {
  std::shared_mutex mu;
  std::map<std::string, const char*, std::less<>> vals;
  std::deque<std::string> pool;  // the values' storage: entries are never freed, so a pointer handed out stays good
  bool loaded = false;
};
Store& store() {
  static Store* s = new Store;  // (never destroyed: kernels' hosts may ask during process exit)
  return *s;
}
void load_locked(Store& st) {
  for (char** e = environ; e && *e; ++e) {
    if (std::strncmp(*e, "CS_", 3) != 0) continue;
    const char* eq = std::strchr(*e, '=');
    if (!eq) continue;
    st.pool.emplace_back(eq + 1);
    st.vals[std::string(*e, (size_t)(eq - *e))] = st.pool.back().c_str();
  }
  st.loaded = true;
}
}  // namespace

const char* cfg(const char* name) {
  Store& st = store();
  {
    std::shared_lock<std::shared_mutex> lk(st.mu);
    if (st.loaded) {
      auto it = st.vals.find(std::string_view(name));
      return it == st.vals.end() ? nullptr : it->second;
    }
  }
  std::unique_lock<std::shared_mutex> lk(st.mu);
  if (!st.loaded) load_locked(st);
  auto it = st.vals.find(std::string_view(name));
  return it == st.vals.end() ? nullptr : it->second;
}

int cfg_int(const char* name, int fallback) {
  const char* e = cfg(name);
  return e ? atoi(e) : fallback;
}

void cfg_set(const char* name, const char* value) {
  Store& st = store();
  std::unique_lock<std::shared_mutex> lk(st.mu);
  if (!st.loaded) load_locked(st);
  if (!value) {
    auto it = st.vals.find(std::string_view(name));
    if (it != st.vals.end()) st.vals.erase(it);
    return;
  }
  st.pool.emplace_back(value);
  st.vals[name] = st.pool.back().c_str();
}

}  // namespace cs
