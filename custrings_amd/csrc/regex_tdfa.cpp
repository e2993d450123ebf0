// Host-side construction of the tagged DFA (layout and semantics: regex_tdfa.h).
// Pure host C++; compiled into the library and into tests/rowemu.
#include "regex_tdfa.h"
#include "cs_config.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <vector>

#include "regex_program.h"

namespace csrx {
// shortest number of characters a match consumes (0: the pattern matches the empty string)
int min_match_chars(const Program& P) {
  // shortest number of consumed characters from any start to END (BFS, 0-1 weights)
  const int n = (int)P.insts.size();
  std::vector<int> dist(n, 1 << 28);
  std::vector<int> q;
  for (int32_t s : P.starts) {
    if (s < 0) break;
    if (s < n) {
      dist[s] = 0;
      q.push_back(s);
    }
  }
  int best = 1 << 28;
  for (size_t h = 0; h < q.size(); ++h) {  // Bellman-Ford style relaxation (tiny graphs)
    int id = q[h];
    const Inst& in = P.insts[id];
    auto relax = [&](int to, int w) {
      if (to < 0 || to >= n) return;
      if (dist[id] + w < dist[to]) {
        dist[to] = dist[id] + w;
        q.push_back(to);
      }
    };
    switch (in.type) {
      case OP_END: best = std::min(best, dist[id]); break;
      case OP_OR:
        relax(in.u1, 0);
        relax(in.u2, 0);
        break;
      case OP_LBRA:
      case OP_RBRA:
      case OP_BOL:
      case OP_EOL:
      case OP_BOW:
      case OP_NBOW: relax(in.u2, 0); break;
      case OP_CHAR:
      case OP_ANY:
      case OP_ANYNL:
      case OP_CCLASS:
      case OP_NCCLASS: relax(in.u2, 1); break;
      default: break;
    }
    if (q.size() > (size_t)n * n * 4 + 64) break;
  }
  return best >= (1 << 28) ? 0 : best;
}
}  // namespace csrx

namespace csrx {
namespace {

using cstd::kMaxSlots;
using csrow::Char;

struct Atom {
  uint64_t sig = 0;  // predicate truth bits
  bool eot = false, nul = false;
  bool isnl = false, isword = false;
};

struct State {
  std::vector<int> kernel;  // ordered, de-duplicated instruction ids
  int cat;                  // bit0 word, bit1 newline, bit2 row start (masked by use)
  int mode;
  bool operator<(const State& o) const {
    if (mode != o.mode) return mode < o.mode;
    if (cat != o.cat) return cat < o.cat;
    return kernel < o.kernel;
  }
};

unsigned cp_next_valid(unsigned cp) {
  ++cp;
  if (cp >= 0xD800 && cp <= 0xDFFF) cp = 0xE000;
  return cp;
}

struct Builder {
  const Program& P;
  csvm::ProgView V;
  std::vector<int> pred_type, pred_arg;
  std::vector<int> inst_pred;  // per instruction: predicate index or -1
  int p_isnl = -1, p_isword = -1;
  bool use_word = false, use_line = false;
  std::vector<Atom> atoms;
  int ascii_atom[128];
  std::vector<std::pair<uint64_t, int>> na_atoms;  // signature -> atom id, non-ASCII chars

  Builder(const Program& p, const std::vector<int32_t>& image, const uint8_t* flags)
      : P(p), V(csvm::make_view(image.data(), flags)) {}

  int add_pred(int type, int arg) {
    for (size_t i = 0; i < pred_type.size(); ++i)
      if (pred_type[i] == type && pred_arg[i] == arg) return (int)i;
    pred_type.push_back(type);
    pred_arg.push_back(arg);
    return (int)pred_type.size() - 1;
  }
  bool eval(int i, Char c) const {
    switch (pred_type[i]) {
      case cstd::P_CHAR: return c == (Char)pred_arg[i];
      case cstd::P_ANY: return c != '\n';
      case cstd::P_ANYNL: return true;
      case cstd::P_CCLASS: return csvm::class_match(V, pred_arg[i], c);
      case cstd::P_NCCLASS: return !csvm::class_match(V, pred_arg[i], c);
      case cstd::P_ISNL: return c == '\n';
      case cstd::P_ISWORD: return csvm::is_word(V, c);
    }
    return false;
  }
  uint64_t signature(Char c) const {
    uint64_t s = 0;
    for (size_t i = 0; i < pred_type.size(); ++i)
      if (eval((int)i, c)) s |= 1ull << i;
    return s;
  }
  int atom_for(uint64_t sig, bool create) {
    for (size_t k = cstd::ATOM_FIRST_CLASS; k < atoms.size(); ++k)
      if (atoms[k].sig == sig) return (int)k;
    if (!create) return -1;
    Atom a;
    a.sig = sig;
    a.isnl = p_isnl >= 0 && ((sig >> p_isnl) & 1);
    a.isword = p_isword >= 0 && ((sig >> p_isword) & 1);
    atoms.push_back(a);
    return (int)atoms.size() - 1;
  }

  bool collect_predicates() {
    const int n = (int)P.insts.size();
    inst_pred.assign(n, -1);
    bool has_big = false;  // a literal or range end among the 4-byte characters
    for (int i = 0; i < n; ++i) {
      const Inst& in = P.insts[i];
      switch (in.type) {
        case OP_CHAR:
          inst_pred[i] = add_pred(cstd::P_CHAR, in.u1);
          has_big |= (uint32_t)in.u1 >= 0xF0000000u;
          break;
        case OP_ANY: inst_pred[i] = add_pred(cstd::P_ANY, 0); break;
        case OP_ANYNL: inst_pred[i] = add_pred(cstd::P_ANYNL, 0); break;
        case OP_CCLASS:
        case OP_NCCLASS:
          if (in.u1 < 0 || in.u1 >= (int)P.classes.size()) return false;
          inst_pred[i] = add_pred(in.type == OP_CCLASS ? cstd::P_CCLASS : cstd::P_NCCLASS, in.u1);
          for (uint32_t r : P.classes[in.u1].ranges) has_big |= r >= 0xF0000000u;
          break;
        case OP_BOL: use_line = true; break;
        case OP_EOL: use_line = true; break;
        case OP_BOW:
        case OP_NBOW: use_word = true; break;
        default: break;
      }
    }
    // '$' needs "next char is a newline", '^' needs "previous char was a newline"
    if (use_line) p_isnl = add_pred(cstd::P_ISNL, 0);
    if (use_word) p_isword = add_pred(cstd::P_ISWORD, 0);
    if (pred_type.size() > 62) return false;
    // atoms
    atoms.resize(2);
    atoms[cstd::ATOM_EOT].eot = true;
    atoms[cstd::ATOM_NUL].eot = true;
    atoms[cstd::ATOM_NUL].nul = true;
    ascii_atom[0] = cstd::ATOM_NUL;
    for (unsigned c = 1; c < 128; ++c) ascii_atom[c] = atom_for(signature(c), true);
    auto add_na = [&](unsigned cp) {
      uint64_t s = signature(csrow::cp_to_packed(cp));
      for (auto& kv : na_atoms)
        if (kv.first == s) return;
      na_atoms.emplace_back(s, atom_for(s, true));
    };
    for (unsigned cp = 0x80; cp < 0x10000; cp = cp_next_valid(cp)) add_na(cp);
    if (has_big) {
      for (unsigned cp = 0x10000; cp < 0x110000; ++cp) add_na(cp);
    } else {
      add_na(0x1F600);
    }
    return atoms.size() <= 250;
  }

  // ---- one step of the list simulator, symbolically --------------------------
  struct Step {
    State next;
    bool stop = false;
    int match = -1;  // -1 none, 0..7 old slot, 15 new
    std::vector<int> origins;
    // capture groups whose LBRA / RBRA lies on the closure path into each surviving thread (bit g-1 = group g),
    // and into the matching thread: the simulator sets that group's begin / end to the current position there
    std::vector<uint32_t> tag_b, tag_e;
    uint32_t match_b = 0, match_e = 0;
  };
  bool step(const State& S, int atom_id, Step& out) const {
    const Atom& a = atoms[atom_id];
    const bool at0 = (S.cat & 4) != 0, pc_nl = (S.cat & 2) != 0, pc_word = (S.cat & 1) != 0;
    const bool cc_zero = a.eot, cc_nl = a.isnl, cc_word = a.isword;
    const int n = (int)P.insts.size();
    std::vector<char> seen(n, 0);
    struct Thr {
      int first, second;  // (inst, origin)
      uint32_t tb, te;    // group tags collected on the closure path
    };
    std::vector<Thr> L;
    std::vector<Thr> stk;  // (inst, -, tags)
    auto gbit = [](int subid) -> uint32_t { return subid >= 1 && subid <= 32 ? 1u << (subid - 1) : 0u; };
    auto closure = [&](int inst, int origin) {
      stk.clear();
      stk.push_back(Thr{inst, 0, 0, 0});
      while (!stk.empty()) {
        const Thr t = stk.back();
        const int id = t.first;
        stk.pop_back();
        if (id < 0 || id >= n) continue;  // malformed program: thread vanishes
        if (seen[id]) continue;
        seen[id] = 1;
        const Inst& in = P.insts[id];
        switch (in.type) {
          case OP_OR:
            stk.push_back(Thr{in.u2, 0, t.tb, t.te});
            stk.push_back(Thr{in.u1, 0, t.tb, t.te});
            break;
          case OP_LBRA: stk.push_back(Thr{in.u2, 0, t.tb | gbit(in.u1), t.te}); break;
          case OP_RBRA: stk.push_back(Thr{in.u2, 0, t.tb, t.te | gbit(in.u1)}); break;
          case OP_BOL:
            if (at0 || ((Char)in.u1 == '^' && pc_nl)) stk.push_back(Thr{in.u2, 0, t.tb, t.te});
            break;
          case OP_EOL:
            if (cc_zero || ((Char)in.u1 == '$' && cc_nl)) stk.push_back(Thr{in.u2, 0, t.tb, t.te});
            break;
          case OP_BOW:
          case OP_NBOW:
            if ((cc_word != pc_word) == (in.type == OP_BOW)) stk.push_back(Thr{in.u2, 0, t.tb, t.te});
            break;
          default: L.push_back(Thr{id, origin, t.tb, t.te}); break;
        }
      }
    };
    for (size_t i = 0; i < S.kernel.size(); ++i) closure(S.kernel[i], (int)i);
    const bool seed = (S.mode == cstd::MODE_RESTART && !(a.eot && !a.nul)) || S.mode == cstd::MODE_SEED_ONCE;
    if (seed)
      for (int32_t s : P.starts) {
        if (s < 0) break;
        closure(s, 15);
      }
    out = Step();
    std::vector<Thr> nk;
    for (auto& t : L) {
      const Inst& in = P.insts[t.first];
      if (in.type == OP_END) {
        out.match = t.second;
        out.match_b = t.tb;
        out.match_e = t.te;
        break;
      }
      int pi = inst_pred[t.first];
      if (pi < 0 || a.eot) continue;  // unknown opcode never advances; nothing advances past the end
      if ((a.sig >> pi) & 1) {
        bool dup = false;
        for (auto& q : nk) dup |= q.first == in.u2;
        if (!dup) nk.push_back(Thr{in.u2, t.second, t.tb, t.te});
      }
    }
    if ((int)nk.size() > cstd::kMaxSlotsWide) return false;  // (five to eight threads: TdfaWide, regex_tdfa.h)
    out.next.mode = (S.mode == cstd::MODE_RESTART && out.match < 0) ? cstd::MODE_RESTART : cstd::MODE_NORESTART;
    out.next.cat = (use_word && a.isword ? 1 : 0) | (use_line && a.isnl ? 2 : 0);
    for (auto& q : nk) {
      out.next.kernel.push_back(q.first);
      out.origins.push_back(q.second);
      out.tag_b.push_back(q.tb);
      out.tag_e.push_back(q.te);
    }
    out.stop = a.eot || (nk.empty() && out.next.mode == cstd::MODE_NORESTART);
    // The reference's search for the next start (regexec.inl:220-258): with no thread alive and a program whose FIRST
    // instruction is a literal character it jumps to that character's next occurrence with custring_view::find -- by length,
    // over anything in between, NUL bytes too; the loop's `while (c && ...)` never sees them.  A NUL byte met with no thread
    // alive therefore ends nothing for such a program (count_re('a') of "a\0a" is 2); met by a live thread it ends the call as before.
    // (a first instruction `^`, multi-line: the jump goes to the byte behind the next line feed at or after the PREVIOUS byte --
    // unless the call stands at byte 0, where nothing jumps; behind a line feed it lands on the NUL itself, which then ends the call)
    bool jumps = false;
    if (P.start_inst >= 0 && (size_t)P.start_inst < P.insts.size()) {
      const Inst& first = P.insts[(size_t)P.start_inst];
      jumps = first.type == OP_CHAR || (first.type == OP_BOL && (Char)first.u1 == '^' && !at0 && !pc_nl);
    }
    if (a.nul && S.kernel.empty() && S.mode == cstd::MODE_RESTART && out.match < 0 && jumps) {
      out.stop = false;
      out.next.kernel.clear();
      out.origins.clear();
      out.tag_b.clear();
      out.tag_e.clear();
      out.next.mode = cstd::MODE_RESTART;
    }
    return true;
  }

  int min_match_chars() const { return csrx::min_match_chars(P); }
};

}  // namespace

std::vector<int32_t> build_tdfa(const Program& prog, const std::vector<int32_t>& image, const uint8_t* flags,
                                std::vector<int32_t>* groups_out) {
  std::vector<int32_t> none;
  if (groups_out) groups_out->clear();
  const int ngroups = std::max(0, std::min(prog.num_groups, 32));
  std::vector<std::vector<uint32_t>> gt(ngroups);  // per group: nstates x natoms tag words (regex_tdfa.h: group_find)
  if (prog.insts.empty() || prog.insts.size() > 4096) return none;
  Builder B(prog, image, flags);
  if (!B.collect_predicates()) return none;
  const int natoms = (int)B.atoms.size();

  std::map<State, int> ids;
  std::vector<State> states;
  auto intern = [&](const State& s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    int id = (int)states.size();
    ids.emplace(s, id);
    states.push_back(s);
    return id;
  };
  // initial states: empty kernel, every (mode, category) the executor can start in
  std::vector<uint32_t> init(3 * 8, 0);
  for (int mode = 0; mode < 3; ++mode)
    for (int cat = 0; cat < 8; ++cat) {
      State s;
      s.mode = mode;
      int c = 0;
      if (cat & 4) c = B.use_line ? 4 : 0;  // row start: previous char is "nothing" (not word, not newline)
      else c = (B.use_word ? (cat & 1) : 0) | (B.use_line ? (cat & 2) : 0);
      s.cat = c;
      init[mode * 8 + cat] = (uint32_t)intern(s);
    }
  std::vector<uint32_t> t2;          // nstates x natoms
  std::vector<uint32_t> act;         // complex origin words
  for (size_t si = 0; si < states.size(); ++si) {
    if (states.size() > (size_t)cstd::kMaxStates) return none;
    t2.resize((si + 1) * natoms, 0);
    for (auto& g : gt) g.resize((si + 1) * natoms, 0);
    for (int a = 0; a < natoms; ++a) {
      Builder::Step st;
      State cur = states[si];  // copy: `states` may grow
      if (!B.step(cur, a, st)) return none;
      uint32_t e = 0;
      if (st.stop) {
        e |= cstd::E_STOP;
      } else {
        e |= (uint32_t)intern(st.next);
      }
      if (st.match >= 0) e |= cstd::E_MATCH | ((uint32_t)st.match << 12);
      // origins: identity prefix + new tail, or complex
      const int m = (int)st.origins.size();
      int k = 0;
      while (k < m && st.origins[k] == k) ++k;
      bool tail_new = true;
      for (int j = k; j < m; ++j) tail_new &= st.origins[j] == 15;
      if (st.stop || m == 0 || k == m) {
        e |= cstd::e_keep_field(15u);
      } else if (tail_new) {
        e |= cstd::e_keep_field((uint32_t)k);
      } else {
        uint32_t og = 0;
        for (int j = 0; j < cstd::kMaxSlotsWide; ++j) og |= (uint32_t)(j < m ? st.origins[j] : 15) << (4 * j);
        size_t idx = std::find(act.begin(), act.end(), og) - act.begin();
        if (idx == act.size()) act.push_back(og);
        if (idx >= 2048) return none;
        e |= cstd::E_COMPLEX | cstd::e_keep_field(15u) | ((uint32_t)idx << 21);
      }
      t2[si * natoms + a] = e;
      for (int g = 0; g < ngroups; ++g) {
        uint32_t w = (((st.match_b >> g) & 1u) << 8) | (((st.match_e >> g) & 1u) << 9);
        for (int j = 0; j < m && j < kMaxSlots; ++j) w |= (((st.tag_b[j] >> g) & 1u) << (2 * j)) | (((st.tag_e[j] >> g) & 1u) << (2 * j + 1));
        gt[g][si * natoms + a] = w;
      }
    }
  }
  const int nstates = (int)states.size();
  if (nstates > cstd::kMaxStates) return none;

  std::vector<int32_t> img(cstd::kHeaderWords, 0);
  auto append = [&](const void* p, size_t words) {
    size_t off = img.size();
    const int32_t* w = (const int32_t*)p;
    img.insert(img.end(), w, w + words);
    return (int32_t)off;
  };
  img[0] = cstd::kMagic;
  img[1] = nstates;
  img[2] = natoms;
  img[3] = (int32_t)B.pred_type.size();
  img[4] = (int32_t)B.na_atoms.size();
  img[5] = (B.use_word ? 1 : 0) | (B.use_line ? 2 : 0);
  img[6] = append(init.data(), init.size());
  std::vector<uint32_t> t1((size_t)nstates * 128);
  for (int s = 0; s < nstates; ++s)
    for (int c = 0; c < 128; ++c) t1[(size_t)s * 128 + c] = t2[(size_t)s * natoms + B.ascii_atom[c]];
  img[7] = append(t1.data(), t1.size());
  img[8] = append(t2.data(), t2.size());
  std::vector<int32_t> preds;
  for (size_t i = 0; i < B.pred_type.size(); ++i) {
    preds.push_back(B.pred_type[i]);
    preds.push_back(B.pred_arg[i]);
  }
  if (preds.empty()) preds.push_back(0);
  img[9] = append(preds.data(), preds.size());
  std::vector<uint32_t> sig;
  for (auto& kv : B.na_atoms) {
    sig.push_back((uint32_t)kv.first);
    sig.push_back((uint32_t)(kv.first >> 32));
    sig.push_back((uint32_t)kv.second);
  }
  img[10] = append(sig.data(), sig.size());
  if (act.empty()) act.push_back(0);
  img[11] = append(act.data(), act.size());
  int maxslots = 0;
  for (auto& s : states) maxslots = std::max(maxslots, (int)s.kernel.size());
  img[12] = maxslots;
  img[13] = B.min_match_chars();
  uint32_t cat[32] = {0};
  for (unsigned c = 1; c < 128; ++c) {
    const Atom& a = B.atoms[B.ascii_atom[c]];
    unsigned v = (B.use_word && a.isword ? 1u : 0u) | (B.use_line && a.isnl ? 2u : 0u);
    cat[c >> 2] |= v << (8 * (c & 3));
  }
  img[14] = append(cat, 32);
  // idle-state skip data (regex_tdfa.h header words 16..28)
  {
    // INIT[RESTART][*] were interned first: ids 0..nskip-1
    uint32_t nskip = 0;
    for (int c = 0; c < 8; ++c) nskip = std::max(nskip, init[cstd::MODE_RESTART * 8 + c] + 1);
    bool ok = true;
    for (uint32_t s = 0; s < nskip; ++s) ok &= states[s].kernel.empty() && states[s].mode == cstd::MODE_RESTART;
    if (!ok) nskip = 0;
    img[16] = (int32_t)nskip;
    for (int c = 0; c < 4; ++c) img[17 + c] = (int32_t)init[cstd::MODE_RESTART * 8 + c];
    uint32_t cand[4] = {~0u, ~0u, ~0u, ~0u}, wordbm[4] = {0, 0, 0, 0};
    for (unsigned c = 1; c < 128; ++c) {
      const Atom& a = B.atoms[B.ascii_atom[c]];
      const unsigned catc = (B.use_word && a.isword ? 1u : 0u) | (B.use_line && a.isnl ? 2u : 0u);
      if (B.use_word && a.isword) wordbm[c >> 5] |= 1u << (c & 31);
      bool skippable = nskip > 0;
      for (uint32_t s = 0; s < nskip && skippable; ++s) {
        uint32_t e = t1[(size_t)s * 128 + c];
        skippable = !(e & (cstd::E_STOP | cstd::E_MATCH | cstd::E_COMPLEX)) && cstd::e_keep(e) == 15u &&
                    (e & cstd::E_STATE) == init[cstd::MODE_RESTART * 8 + catc];
      }
      if (skippable) cand[c >> 5] &= ~(1u << (c & 31));
    }
    for (int k = 0; k < 4; ++k) {
      img[21 + k] = (int32_t)cand[k];
      img[25 + k] = (int32_t)wordbm[k];
    }
    // two ranges covering the candidate bytes 1..127: split at the widest gap
    std::vector<int> cs;
    for (int c = 1; c < 128; ++c)
      if ((cand[c >> 5] >> (c & 31)) & 1u) cs.push_back(c);
    int lo1 = 1, hi1 = 0, lo2 = 1, hi2 = 0;  // empty ranges
    if (!cs.empty()) {
      size_t cut = 0;
      int gap = 0;
      for (size_t i = 0; i + 1 < cs.size(); ++i)
        if (cs[i + 1] - cs[i] > gap) {
          gap = cs[i + 1] - cs[i];
          cut = i;
        }
      lo1 = cs.front();
      if (gap > 1) {
        hi1 = cs[cut];
        lo2 = cs[cut + 1];
        hi2 = cs.back();
      } else {
        hi1 = cs.back();
      }
    }
    img[29] = lo1 | (hi1 << 8);
    img[30] = lo2 | (hi2 << 8);
  }
  // EXIT marks (regex_tdfa.h): non-COMPLEX transitions into an idle state, in T1 and T2
  {
    const uint32_t nskip = (uint32_t)img[16];
    auto mark = [&](int32_t off, size_t count) {
      for (size_t i = 0; i < count; ++i) {
        uint32_t e = (uint32_t)img[off + i];
        if (!(e & (cstd::E_COMPLEX | cstd::E_STOP)) && (e & cstd::E_STATE) < nskip) img[off + i] = (int32_t)(e | cstd::E_EXIT);
      }
    };
    mark(img[7], (size_t)nstates * 128);
    mark(img[8], (size_t)nstates * natoms);
  }
  // Unit decomposition for the replace kernels (regex_tdfa.h header word 31).  A KILLER byte sends every state to
  // an idle state or stops the automaton, starting nothing: whatever the scan did before it, it continues
  // behind it exactly as a fresh scan would.  A row then falls into independent UNITS -- maximal runs of
  // non-killer bytes that hold a candidate byte -- which any lane may scan on its own.  The device finds the
  // runs from two per-byte bitmaps, "candidate" (the two header ranges, a superset) and "equals x", so the
  // decomposition is offered when every non-killer byte is inside the candidate ranges or is the one byte x;
  // when no match exists without an x, runs that hold none are not scanned at all (for the dotted-quad
  // pattern: x = '.', and a run of digits alone, a status code say, costs nothing).
  {
    const uint32_t nskip = (uint32_t)img[16];
    const uint32_t* T1 = (const uint32_t*)(img.data() + img[7]);
    const uint32_t* T2 = (const uint32_t*)(img.data() + img[8]);
    const int lo1 = img[29] & 255, hi1 = (img[29] >> 8) & 255, lo2 = img[30] & 255, hi2 = (img[30] >> 8) & 255;
    auto in_ranges = [&](int c) { return (c >= lo1 && c <= hi1) || (c >= lo2 && c <= hi2); };
    int32_t word = 0;
    if (nskip > 0 && maxslots <= 4 && B.min_match_chars() >= 1) {
      int extras = 0, x = 0;
      for (int c = 1; c < 128; ++c) {
        bool killer = true;
        for (int st = 0; st < nstates && killer; ++st) {
          const uint32_t e = T1[(size_t)st * 128 + c];
          killer = !(e & cstd::E_COMPLEX) && cstd::e_keep(e) == 15u && ((e & cstd::E_STOP) || (e & cstd::E_STATE) < nskip);
        }
        if (!killer && !in_ranges(c)) {
          ++extras;
          x = c;
        }
      }
      if (extras <= 1) {
        bool required = false;
        if (extras == 1) {
          // is a match reachable from the start states without ever consuming x?
          std::vector<char> seen((size_t)nstates, 0);
          std::vector<int> todo;
          for (int c = 0; c < 8; ++c) {
            const int st = (int)init[cstd::MODE_RESTART * 8 + c];
            if (!seen[(size_t)st]) {
              seen[(size_t)st] = 1;
              todo.push_back(st);
            }
          }
          bool match = false;
          while (!todo.empty() && !match) {
            const int st = todo.back();
            todo.pop_back();
            if (T2[(size_t)st * natoms + cstd::ATOM_EOT] & cstd::E_MATCH) match = true;
            for (int c = 1; c < 128 && !match; ++c) {
              if (c == x) continue;
              const uint32_t e = T1[(size_t)st * 128 + c];
              if (e & cstd::E_MATCH) match = true;
              const int nx = (int)(e & cstd::E_STATE);
              if (!(e & cstd::E_STOP) && nx < nstates && !seen[(size_t)nx]) {
                seen[(size_t)nx] = 1;
                todo.push_back(nx);
              }
            }
          }
          required = !match;
        }
        word = 1 | (x << 8) | (required ? 1 << 16 : 0);
        // bit 17: a non-ASCII character can only kill -- no atom matches one (every non-ASCII signature is empty), no
        // anchors, no \b -- exactly as the end of the row does: the unit route may take tiles that hold such bytes,
        // a unit's scan ending where the unit ends (cs_regex.hip: reclassify_high)
        bool only_kills = !B.use_word && !B.use_line;
        for (const auto& kv : B.na_atoms) only_kills = only_kills && kv.first == 0;
        if (only_kills) word |= 1 << 17;
        // bits 18-23: the same for the non-ASCII characters whose unicode flags miss a mask -- a pattern of ASCII literals
        // and of classes made of ASCII ranges and \w / \s / \d (not negated): a non-ASCII character matches one of its
        // atoms only through a builtin, i.e. when flags[cp] & mask != 0 (regex_vm.h: class_match; \w: 15, \s: 16, \d: 4).
        // The kernels then check the characters of a tile against the mask (a column of accented text holds no digit
        // outside ASCII: `\d+\.\d+\.\d+\.\d+` stays on the unit route there).
        if (!only_kills && !B.use_word && !B.use_line) {
          bool ok = true;
          unsigned fmask = 0;
          for (size_t i = 0; i < B.pred_type.size() && ok; ++i) {
            if (B.pred_type[i] == cstd::P_CHAR) {
              ok = (uint32_t)B.pred_arg[i] < 128u;
            } else if (B.pred_type[i] == cstd::P_CCLASS) {
              const CharClass& cc = prog.classes[B.pred_arg[i]];
              for (uint32_t r : cc.ranges) ok = ok && r < 128u;
              ok = ok && (cc.builtins & ~7) == 0;
              fmask |= ((cc.builtins & 1) ? 15u : 0u) | ((cc.builtins & 2) ? 16u : 0u) | ((cc.builtins & 4) ? 4u : 0u);
            } else {
              ok = false;  // `.`, a negated class
            }
          }
          if (ok && fmask) word |= (1 << 18) | (int32_t)(fmask << 19);
        }
      }
    }
    // bit 25: the bytes a column's VIRTUAL ROWS are cut behind -- space, tab, LF, CR (cs_virtual.hip: a long row cut into
    // pieces of at most 92 bytes so that the kernels on 96-bit masks take it) -- are SAFE CUTS for this program: whatever
    // state consumes one, nothing is kept, the automaton does not stop, and the idle state it lands in behaves exactly as
    // the start of a row does (the same row of byte transitions and of end-of-text / assertion atoms: a program with `^`
    // or a `\b` that tells a row's start from "behind a space" fails here).  A match can then neither span a cut nor tell
    // the piece from the row, so an op on the pieces is the op on the rows.
    if (nskip > 0 && maxslots <= 4 && B.min_match_chars() >= 1) {
      const uint32_t a0 = init[cstd::MODE_RESTART * 8 + 4];  // the start of a row
      auto same_rows = [&](uint32_t u, uint32_t v) {
        if (u == v) return true;
        for (int c = 0; c < 128; ++c)
          if (T1[(size_t)u * 128 + c] != T1[(size_t)v * 128 + c]) return false;
        for (int k = 0; k < natoms; ++k)
          if (T2[(size_t)u * natoms + k] != T2[(size_t)v * natoms + k]) return false;
        return true;
      };
      bool safe = a0 < nskip;
      for (int c : {9, 10, 13, 32}) {
        // (1) a killer: whichever state consumes it, nothing is kept; the scan stops (the seeded / no-restart states: a later
        // seed starts from an init state again) or lands in an idle state that behaves as the start of a row does
        for (int st = 0; st < nstates && safe; ++st) {
          const uint32_t e = T1[(size_t)st * 128 + c];
          safe = !(e & cstd::E_COMPLEX) && cstd::e_keep(e) == 15u && ((e & cstd::E_STOP) || ((e & cstd::E_STATE) < nskip && same_rows(e & cstd::E_STATE, a0)));
        }
        // (2) a scan seeded right behind it starts as one seeded at a row's first byte does (regex_tdfa.h: prev_cat -- category
        // 4 is the row's start, a space's is 0, a line feed's 2), in every mode
        const int k = c == 10 ? 2 : 0;
        for (int m = 0; m < 3 && safe; ++m) safe = same_rows(init[m * 8 + k], init[m * 8 + 4]);
      }
      if (safe) word |= 1 << 25;
    }
    img[31] = word;
    // The CHAIN form (regex_tdfa.h: chain_match), offered beside the unit decomposition: the program is a straight line
    // of single-character items -- a literal or a class, taken once, in a greedy `+` loop or a counted number of times
    // (`{m,n}`: the compiler's m copies and n - m nested optional ones, regex_compile.cpp: expand_counted), brackets
    // ignored -- every item's ASCII members are exactly the candidate ranges (class R) or exactly the byte x, neighbours
    // differ, the first is R, and the last is repeated when it is R as well.  Then a plain-ASCII row's matches follow from
    // its two per-byte masks by integer arithmetic alone.
    // A chain may end in a SUFFIX of up to four literal ASCII bytes outside R (`\d+\.\d+\.\d+\.\d+ `): the chain part's
    // ends are then filtered by a byte compare (chain_suffix_filter).  Such a pattern usually has no unit decomposition
    // (two non-killer bytes outside the ranges); x is then the first literal of the line that is not in R, and header word
    // 31 carries it WITHOUT bit 0 (the kernels stage the "equals x" bitmap from it; the unit route stays off).
    // A `\b` may open the line and close it ([30] bits 24 / 25) when the class beside it holds letters and digits only: the
    // boundary is then "the byte on the other side is no word character", a byte test per match (chain_match).
    if (nskip > 0 && maxslots <= 4 && B.min_match_chars() >= 1 && !B.use_line && !cs::cfg("CS_NO_CHAIN")) {
      int x = (word >> 8) & 127;
      const bool x_free = !(word & 1) && !cs::cfg("CS_NO_CHAIN_SUFFIX");  // (no unit decomposition: the chain picks its own x)
      const bool counted_ok = !cs::cfg("CS_NO_CHAIN_COUNTED");  // (off: `{m,n}` items and `\b` keep the pattern off the chain form)
      uint32_t items = 0;
      unsigned long long crep = 0;
      int ni = 0;
      size_t seen = 0;
      bool ok = true, lead_b = false, trail_b = false;
      int pc = prog.start_inst, prev_cls = -1;
      int g_lo[5] = {-1, -1, -1, -1, -1}, g_hi[5] = {-1, -1, -1, -1, -1};  // capture groups 1..4: the items they span
      bool groups_ok = true;
      uint32_t sfx = 0;
      int slen = 0;
      auto inst_at = [&](int at) -> const Inst* { return at >= 0 && (size_t)at < prog.insts.size() ? &prog.insts[(size_t)at] : nullptr; };
      auto same_atom = [](const Inst& a, const Inst& b) { return a.type == b.type && a.u1 == b.u1 && (a.type == OP_CHAR || a.type == OP_CCLASS); };
      auto word_byte = [](int c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
      bool r_is_word = true;  // every byte of R a letter or a digit?
      for (int c = 1; c < 128; ++c) r_is_word = r_is_word && (!in_ranges(c) || word_byte(c));
      while (ok) {
        if (pc < 0 || (size_t)pc >= prog.insts.size() || seen > prog.insts.size()) {
          ok = false;
          break;
        }
        const Inst& in = prog.insts[(size_t)pc];
        ++seen;
        if (in.type == OP_END) break;
        if (in.type == OP_LBRA || in.type == OP_RBRA) {
          if (slen > 0) groups_ok = false;  // (brackets inside the suffix: no group map)
          else if (in.u1 >= 1 && in.u1 <= 4) (in.type == OP_LBRA ? g_lo : g_hi)[in.u1] = ni;
          else groups_ok = false;
          pc = in.u2;
          continue;
        }
        if (in.type == OP_BOW && counted_ok && !trail_b && slen == 0 && (ni > 0 || !lead_b)) {
          (ni == 0 ? lead_b : trail_b) = true;
          pc = in.u2;
          continue;
        }
        if ((in.type != OP_CHAR && in.type != OP_CCLASS) || trail_b) {  // (behind the closing `\b`: brackets and the end only)
          ok = false;
          break;
        }
        const bool lit_off_r = in.type == OP_CHAR && (uint32_t)in.u1 >= 1u && (uint32_t)in.u1 < 128u && !in_ranges((int)in.u1);
        if (slen > 0) {  // inside the suffix: literals outside R only
          if (!lit_off_r || slen == 4 || cs::cfg("CS_NO_CHAIN_SUFFIX")) {
            ok = false;
            break;
          }
          sfx |= (uint32_t)in.u1 << (8 * slen);
          ++slen;
          pc = in.u2;
          continue;
        }
        if (x == 0 && x_free && lit_off_r && ni > 0) x = (int)in.u1;
        // the item's ASCII members against R and against {x}
        bool is_r = true, is_x = x != 0;
        for (int c = 1; c < 128; ++c) {
          const bool m = in.type == OP_CHAR ? (uint32_t)in.u1 == (uint32_t)c : csvm::class_match(B.V, in.u1, (csvm::Char)c);
          is_r = is_r && m == in_ranges(c);
          is_x = is_x && m == (c == x);
        }
        if (in.type == OP_CHAR && (uint32_t)in.u1 >= 128u) is_r = is_x = false;
        const int cls = is_r ? 0 : (is_x ? 1 : -1);
        if (cls < 0 && lit_off_r && ni > 0 && !cs::cfg("CS_NO_CHAIN_SUFFIX")) {  // the suffix begins
          sfx = (uint32_t)in.u1;
          slen = 1;
          pc = in.u2;
          continue;
        }
        if (cls < 0 || cls == prev_cls || (ni == 0 && cls != 0) || ni == 8) {
          ok = false;
          break;
        }
        // how often: copies of the item in a row, then a loop back to the last one (`+`, `{m,}`) or nested optional copies
        int least = 1, most = 1, last = pc, next = in.u2;
        while (counted_ok) {
          const Inst* nx = inst_at(next);
          if (!nx || !same_atom(*nx, in) || least == 15) break;
          ++least;
          ++seen;
          last = next;
          next = nx->u2;
        }
        most = least;
        if (const Inst* o = inst_at(next); o && o->type == OP_OR && o->u1 == last) {
          most = 0;  // (the preferred branch of the OR goes back to the item: a greedy loop)
          ++seen;
          next = o->u2;
        } else if (o && o->type == OP_OR && counted_ok && inst_at(o->u1) && same_atom(*inst_at(o->u1), in)) {
          // OR(copy -> OR(copy -> ... -> exit, exit), exit): every copy optional, each only behind the one before
          const int exit = o->u2;
          const Inst* cur = o;
          while (ok) {
            const Inst* copy = inst_at(cur->u1);
            if (!copy || !same_atom(*copy, in) || cur->u2 != exit || most == 15) {
              ok = false;
              break;
            }
            ++most;
            seen += 2;
            if (copy->u2 == exit) break;
            cur = inst_at(copy->u2);
            if (!cur || cur->type != OP_OR) ok = false;
          }
          if (!ok) break;
          next = exit;
        }
        items |= (uint32_t)(cls | (most == 0 ? 2 : 0)) << (2 * ni);
        crep |= (unsigned long long)(uint32_t)(least | (most << 4)) << (8 * ni);
        ++ni;
        prev_cls = cls;
        pc = next;
      }
      // (every instruction on the line: an alternation or an optional part would leave some unvisited)
      ok = ok && ni > 0 && seen == prog.insts.size();
      if (ok) {
        const uint32_t first = (uint32_t)(crep & 255u), tail = (uint32_t)(crep >> (8 * (ni - 1))) & 255u;
        const bool tail_r = ((items >> (2 * (ni - 1))) & 1u) == 0;
        // R ... R: the tail takes its whole run (the scan resumes behind a match, and never inside a first run)
        if (tail_r && (tail >> 4) == 1 && !trail_b) ok = false;
        // a bounded run in front can be entered in its middle unless a `\b` pins the start; one at the end must be followed
        // by something its class does not hold
        if ((first >> 4) > 1 && !lead_b) ok = false;
        if ((tail >> 4) > 1 && !trail_b && slen == 0) ok = false;
        // `\b` beside a class of word characters only (and the closing one not behind a suffix)
        if (lead_b && !r_is_word) ok = false;
        if (trail_b && (slen > 0 || !(tail_r ? r_is_word : word_byte(x)))) ok = false;
      }
      // (without a unit decomposition the chain is only worth offering when it brought a suffix or its own x: a plain
      // chain of a pattern whose decomposition failed for another reason stays where it was)
      if (ok && !(word & 1) && slen == 0 && ((word >> 8) & 127) == x) ok = false;
      if (ok) {
        img[29] |= (int32_t)(items << 16);
        img[30] |= (int32_t)((uint32_t)ni << 16) | (lead_b ? 1 << 24 : 0) | (trail_b ? 1 << 25 : 0);
        // bit 26: the general (counted) form of the arithmetic -- some item is neither single nor `+`, or a `\b` stands at an end
        bool general = lead_b || trail_b;
        for (int k = 0; k < ni; ++k) general = general || (((crep >> (8 * k)) & 255u) != 0x11u && ((crep >> (8 * k)) & 255u) != 0x01u);
        if (general) img[30] |= 1 << 26;
        if (!(word & 1) && x != ((word >> 8) & 127)) img[31] = word | (x << 8);
        // the image's tail: the repetition counts (two words), then the suffix, then the group map (make_view reads them back
        // from the end)
        img.push_back((int32_t)(uint32_t)crep);
        img.push_back((int32_t)(uint32_t)(crep >> 32));
        if (slen > 0) {
          img[30] |= (int32_t)((uint32_t)slen << 21);
          img.push_back((int32_t)sfx);
        }
        // the capture groups of a chain are runs of items: [30] bit 20 = the image's LAST word holds, a byte per group
        // (1..4), the first item of the group in its low nibble and the item behind its last one in the high nibble (the
        // backrefs kernel reads a match's group ranges off the item boundaries)
        uint32_t gmap = 0;
        for (int g = 1; g <= 4 && groups_ok; ++g) {
          if (g <= prog.num_groups) {
            if (g_lo[g] < 0 || g_hi[g] < g_lo[g]) groups_ok = false;
            else gmap |= (uint32_t)(g_lo[g] | (g_hi[g] << 4)) << (8 * (g - 1));
          }
        }
        if (groups_ok && prog.num_groups >= 1 && prog.num_groups <= 4) {
          img[30] |= 1 << 20;
          img.push_back((int32_t)gmap);
        }
      }
    }
  }
  img[15] = (int32_t)img.size();
  if (groups_out && ngroups > 0 && maxslots <= kMaxSlots) {  // (the tag words hold four slots: wider programs track no groups on the DFA)
    // group-tag image: [0] groups [1] nstates [2] natoms [3] words per table, [4..35] the atom of each ASCII byte
    // (four per word), then one nstates x natoms table per group
    std::vector<int32_t>& G = *groups_out;
    G.assign(4 + 32, 0);
    G[0] = ngroups;
    G[1] = nstates;
    G[2] = natoms;
    G[3] = nstates * natoms;
    for (unsigned c = 0; c < 128; ++c) G[4 + (c >> 2)] |= (int32_t)((uint32_t)(B.ascii_atom[c] & 255) << (8 * (c & 3)));
    for (int g = 0; g < ngroups; ++g) G.insert(G.end(), gt[g].begin(), gt[g].begin() + (size_t)nstates * natoms);
    // then, for the backward resolution of a match's groups (regex_tdfa.h: group_find_back), one table per batch of
    // four groups with the slot tags of the whole batch in ONE word: bit 8j + 2q (+1) = slot j passes the begin (end)
    // bracket of the batch's q-th group
    for (int g0 = 0; g0 < ngroups; g0 += 4) {
      std::vector<int32_t> pt((size_t)nstates * natoms, 0);
      for (int q = 0; q < 4 && g0 + q < ngroups; ++q)
        for (size_t i = 0; i < (size_t)nstates * natoms; ++i) {
          const uint32_t w = (uint32_t)gt[g0 + q][i];
          uint32_t o = 0;
          for (int j = 0; j < kMaxSlots; ++j) o |= ((w >> (2 * j)) & 3u) << (8 * j + 2 * q);
          pt[i] = (int32_t)((uint32_t)pt[i] | o);
        }
      G.insert(G.end(), pt.begin(), pt.end());
    }
  }
  return img;
}

}  // namespace csrx
