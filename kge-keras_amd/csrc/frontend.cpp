// Host-side data front-end of the evaluation path: the filter index of `validate` (KGEkeras/utils.py:201-205 builds
// filter_h[(p,o)] / filter_t[(s,p)] as dicts of sets; gen_tail_data / gen_head_data (:165-196) then remove those ids from
// range(num_entities) per query).  Here: sort + binary search, output in the COO form kge_rank_counts consumes.
#include <algorithm>
#include <cstring>
#include <vector>

#include "kge_host.hpp"

namespace {
struct Key {
  uint64_t key;  // (a << 32) | b : tail side (s,p) ; head side (p,o)
  int32_t id;    // the entity to exclude: o (tail) or s (head)
  bool operator<(const Key& r) const { return key < r.key || (key == r.key && id < r.id); }
  bool operator==(const Key& r) const { return key == r.key && id == r.id; }
};
}  // namespace

extern "C" {

// Pass 1 (out_q == NULL): returns the number of COO entries in *n_out.  Pass 2: fills out_q/out_id (capacity *n_out).
// KGE_EINVAL when a query's target is itself filtered or a filtered id is outside [0, num_entities): the reference
// raises ValueError there (list.remove of an absent element, utils.py:168-171 / :184-188).
int kge_filter_coo_host(const int32_t* test, int64_t nq, int32_t side, const int32_t* filt, int64_t n_filt, int64_t num_entities,
                        int32_t* out_q, int32_t* out_id, int64_t* n_out) {
  if ((!test && nq) || (!filt && n_filt) || !n_out || nq < 0 || n_filt < 0 || (side != 0 && side != 1))
    return kge::fail(KGE_EINVAL, "kge_filter_coo_host: bad arguments");
  std::vector<Key> keys((size_t)n_filt);
  for (int64_t i = 0; i < n_filt; ++i) {
    const int32_t s = filt[3 * i], p = filt[3 * i + 1], o = filt[3 * i + 2];
    keys[i] = side == 0 ? Key{((uint64_t)(uint32_t)s << 32) | (uint32_t)p, o} : Key{((uint64_t)(uint32_t)p << 32) | (uint32_t)o, s};
  }
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());  // the reference's filters are sets
  int64_t n = 0;
  const int64_t cap = out_q ? *n_out : 0;
  for (int64_t q = 0; q < nq; ++q) {
    const int32_t s = test[3 * q], p = test[3 * q + 1], o = test[3 * q + 2];
    const int32_t target = side == 0 ? o : s;
    if (target < 0 || target >= num_entities) return kge::fail(KGE_EINVAL, "list.remove(x): x not in list (query %lld: target outside range(num_entities))", (long long)q);
    const uint64_t k = side == 0 ? (((uint64_t)(uint32_t)s << 32) | (uint32_t)p) : (((uint64_t)(uint32_t)p << 32) | (uint32_t)o);
    auto it = std::lower_bound(keys.begin(), keys.end(), Key{k, INT32_MIN});
    for (; it != keys.end() && it->key == k; ++it) {
      if (it->id == target || it->id < 0 || it->id >= num_entities)
        return kge::fail(KGE_EINVAL, "list.remove(x): x not in list (query %lld: the evaluated triple is in the filter, or a filtered id is out of range)", (long long)q);
      if (out_q) {
        if (n >= cap) return kge::fail(KGE_EWORKSPACE, "kge_filter_coo_host: output capacity %lld too small", (long long)cap);
        out_q[n] = (int32_t)q;
        out_id[n] = it->id;
      }
      ++n;
    }
  }
  *n_out = n;
  return KGE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Dataset loader: `load_kg` (KGEkeras/utils.py:106-112: every line stripped and split on whitespace) for all splits at once, plus
// the id mapping of the reference's script (tests.py:102-117) made deterministic: the script enumerates Python sets of strings,
// so its ids change with PYTHONHASHSEED; here an id is the rank of the string among the sorted distinct strings of all splits
// (entities: subjects and objects; relations: predicates) -- Python's sorted() on str orders by code point, which for UTF-8 is
// the order of the bytes.
#include <cstdio>
#include <string>
#include <unordered_map>

namespace {

// Python's str.split() separators (str.isspace): ASCII \t \n \v \f \r, 0x1c-0x1f, space; U+0085, U+00A0, U+1680, U+2000-U+200A,
// U+2028, U+2029, U+202F, U+205F, U+3000.  Returns the byte length of the whitespace character at p (0: not whitespace).
inline int space_len(const unsigned char* p, const unsigned char* end) {
  const unsigned char c = *p;
  if (c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f)) return 1;
  if (c == 0xc2 && p + 1 < end && (p[1] == 0x85 || p[1] == 0xa0)) return 2;
  if (p + 2 < end) {
    if (c == 0xe1 && p[1] == 0x9a && p[2] == 0x80) return 3;
    if (c == 0xe2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8a) || p[2] == 0xa8 || p[2] == 0xa9 || p[2] == 0xaf)) return 3;
    if (c == 0xe2 && p[1] == 0x81 && p[2] == 0x9f) return 3;
    if (c == 0xe3 && p[1] == 0x80 && p[2] == 0x80) return 3;
  }
  return 0;
}

struct Kg {
  std::vector<std::vector<std::string>> tok;  // per split: 3 tokens per row, flattened
  std::vector<std::string> ents, rels;        // sorted distinct
  std::vector<std::vector<int32_t>> ids;      // per split (rows, 3)
};

int read_split(const char* path, std::vector<std::string>* out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return kge::fail(KGE_EINVAL, "[Errno 2] No such file or directory: '%s'", path);
  std::string data;
  char buf[1 << 16];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, k);
  std::fclose(f);
  const unsigned char* p = reinterpret_cast<const unsigned char*>(data.data());
  const unsigned char* end = p + data.size();
  int64_t line = 1;
  while (p < end) {
    // one line: up to \n, \r\n or \r (text mode with universal newlines)
    const unsigned char* e = p;
    while (e < end && *e != '\n' && *e != '\r') ++e;
    int ntok = 0;
    const unsigned char* q = p;
    while (q < e) {
      int sl;
      while (q < e && (sl = space_len(q, e)) > 0) q += sl;
      if (q >= e) break;
      const unsigned char* t0 = q;
      while (q < e && space_len(q, e) == 0) ++q;
      out->emplace_back(reinterpret_cast<const char*>(t0), (size_t)(q - t0));
      ++ntok;
    }
    if (ntok != 3)  // the reference unpacks every row as `for a, b, c in train` (tests.py:102): anything else is a ValueError there
      return kge::fail(KGE_EINVAL, "%s line %lld: %s values to unpack (expected 3, got %d)", path, (long long)line, ntok < 3 ? "not enough" : "too many", ntok);
    p = e;
    if (p < end) { if (*p == '\r' && p + 1 < end && p[1] == '\n') ++p; ++p; }
    ++line;
  }
  return KGE_OK;
}

}  // namespace

extern "C" {

int kge_kg_open(const char* const* paths_host, int32_t n_splits, void** handle) {
  if (!paths_host || n_splits <= 0 || !handle) return kge::fail(KGE_EINVAL, "kge_kg_open: bad arguments");
  Kg* kg = new Kg();
  kg->tok.resize(n_splits);
  for (int i = 0; i < n_splits; ++i)
    if (int rc = read_split(paths_host[i], &kg->tok[i])) { delete kg; return rc; }
  for (const auto& t : kg->tok)
    for (size_t r = 0; r + 2 < t.size(); r += 3) { kg->ents.push_back(t[r]); kg->rels.push_back(t[r + 1]); kg->ents.push_back(t[r + 2]); }
  auto uniq = [](std::vector<std::string>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
  uniq(kg->ents);
  uniq(kg->rels);
  if (kg->ents.size() > 0x7FFFFFFFull || kg->rels.size() > 0x7FFFFFFFull) { delete kg; return kge::fail(KGE_EUNSUPPORTED, "kge_kg_open: more than 2^31 distinct names"); }
  std::unordered_map<std::string, int32_t> em, rm;
  em.reserve(kg->ents.size() * 2);
  rm.reserve(kg->rels.size() * 2);
  for (size_t i = 0; i < kg->ents.size(); ++i) em.emplace(kg->ents[i], (int32_t)i);
  for (size_t i = 0; i < kg->rels.size(); ++i) rm.emplace(kg->rels[i], (int32_t)i);
  kg->ids.resize(n_splits);
  for (int i = 0; i < n_splits; ++i) {
    const auto& t = kg->tok[i];
    auto& d = kg->ids[i];
    d.resize(t.size());
    for (size_t r = 0; r < t.size(); r += 3) { d[r] = em[t[r]]; d[r + 1] = rm[t[r + 1]]; d[r + 2] = em[t[r + 2]]; }
    std::vector<std::string>().swap(kg->tok[i]);  // the strings of the rows are no longer needed
  }
  *handle = kg;
  return KGE_OK;
}

int kge_kg_sizes(void* handle, int64_t* n_entities, int64_t* n_relations, int64_t* rows_per_split_host) {
  Kg* kg = static_cast<Kg*>(handle);
  if (!kg || !n_entities || !n_relations) return kge::fail(KGE_EINVAL, "kge_kg_sizes: bad arguments");
  *n_entities = (int64_t)kg->ents.size();
  *n_relations = (int64_t)kg->rels.size();
  if (rows_per_split_host)
    for (size_t i = 0; i < kg->ids.size(); ++i) rows_per_split_host[i] = (int64_t)kg->ids[i].size() / 3;
  return KGE_OK;
}

int kge_kg_triples(void* handle, int32_t split, int32_t* out_host) {
  Kg* kg = static_cast<Kg*>(handle);
  if (!kg || split < 0 || (size_t)split >= kg->ids.size() || (!out_host && !kg->ids[split].empty())) return kge::fail(KGE_EINVAL, "kge_kg_triples: bad arguments");
  if (!kg->ids[split].empty()) std::memcpy(out_host, kg->ids[split].data(), kg->ids[split].size() * sizeof(int32_t));
  return KGE_OK;
}

int kge_kg_names(void* handle, int32_t which, char* buf_host, int64_t buf_bytes, int64_t* offsets_host) {
  Kg* kg = static_cast<Kg*>(handle);
  if (!kg || (which != 0 && which != 1) || !offsets_host) return kge::fail(KGE_EINVAL, "kge_kg_names: bad arguments");
  const auto& v = which == 0 ? kg->ents : kg->rels;
  int64_t off = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    offsets_host[i] = off;
    if (buf_host) {
      if (off + (int64_t)v[i].size() > buf_bytes) return kge::fail(KGE_EWORKSPACE, "kge_kg_names: buffer of %lld bytes too small", (long long)buf_bytes);
      std::memcpy(buf_host + off, v[i].data(), v[i].size());
    }
    off += (int64_t)v[i].size();
  }
  offsets_host[v.size()] = off;
  return KGE_OK;
}

int kge_kg_close(void* handle) {
  delete static_cast<Kg*>(handle);
  return KGE_OK;
}

}  // extern "C"
