// Process configuration: the CS_* switches (measurement aids, route overrides for tests, opt-in experiments).
// The environment is read ONCE -- at the first question, in practice inside cs_init -- into an immutable-by-default
// table; the dispatch paths ask the table, never getenv (which is neither cheap nor safe against a concurrent setenv).
// A switch changes at run time only through cs_config_set (C ABI; tests and tools), under the table's lock.
#pragma once
namespace cs {
// the switch's value, or nullptr when it is not set (the pointer stays valid for the life of the process)
const char* cfg(const char* name);
void cfg_set(const char* name, const char* value);  // value == nullptr: unset
// the switch as an integer, `fallback` when it is not set -- ONE look-up (a switch may be unset from another thread between two)
int cfg_int(const char* name, int fallback);
}  // namespace cs
