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
