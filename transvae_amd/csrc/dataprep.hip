// Data pipeline either side of the train step:
//   tvae_encode_smiles  HOST: SMILES text -> rows of token ids (tokenizer + encode_smiles + the row layout of
//                       vae_data_gen, reference transvae/tvae_util.py:97-122, transvae/data.py:26-32), threaded
//   tvae_gather_batch   DEVICE: rows of a uint8 dataset resident in HBM -> the float32 (B, L+1) batch the
//                       reference's DataLoader would have collated (transvae/trans_models.py:176-183, 211)
#include <hip/hip_runtime.h>

#include <cstring>
#include <thread>
#include <vector>

#include "../../include/tvae_kernels.h"
#include "common.h"

namespace {

// ---------------------------------------------------------------------------- tokenizer (host)
// The reference's regex, alternative by alternative (tvae_util.py:99):
//   \[[^\]]+]            '[' , one or more bytes that are not ']' , ']'
//   Br? Cl?              two-letter halogens before the one-letter fallback
//   N O S P F I b c n o s p ( ) . = # - + \ / _ : ~ @ ? > * $      single bytes
//   %[0-9]{2}            ring closure with two digits
//   [0-9]
// findall skips what matches nothing and the reference then asserts that the tokens join back to the input,
// so any byte outside these forms is an error (TVAE_TOK_UNMATCHED), not something to drop.
struct Vocab {
  int16_t single[256];                       // id of the one-byte token, -1 if not in the vocabulary
  struct Multi { const char* s; int len; int id; };
  std::vector<Multi> multi;                  // tokens of two or more bytes (a handful)
};

inline bool is_single_token(unsigned char c) {
  switch (c) {
    case 'B': case 'C': case 'N': case 'O': case 'S': case 'P': case 'F': case 'I':
    case 'b': case 'c': case 'n': case 'o': case 's': case 'p':
    case '(': case ')': case '.': case '=': case '#': case '-': case '+': case '\\': case '/': case '_':
    case ':': case '~': case '@': case '?': case '>': case '*': case '$':
      return true;
    default:
      return c >= '0' && c <= '9';
  }
}

// length of the token that starts at s[0] (n bytes left), 0 if no alternative matches there
inline int token_length(const char* s, int64_t n) {
  const unsigned char c = (unsigned char)s[0];
  if (c == '[') {
    if (n < 3 || s[1] == ']') return 0;
    const void* close = memchr(s + 2, ']', (size_t)(n - 2));
    return close ? (int)((const char*)close - s) + 1 : 0;
  }
  if (c == 'B') return (n > 1 && s[1] == 'r') ? 2 : 1;
  if (c == 'C') return (n > 1 && s[1] == 'l') ? 2 : 1;
  if (c == '%') return (n > 2 && s[1] >= '0' && s[1] <= '9' && s[2] >= '0' && s[2] <= '9') ? 3 : 0;
  return is_single_token(c) ? 1 : 0;
}

inline int lookup(const Vocab& v, const char* s, int len) {
  if (len == 1) return v.single[(unsigned char)s[0]];
  for (const auto& m : v.multi)
    if (m.len == len && memcmp(m.s, s, (size_t)len) == 0) return m.id;
  return -1;
}

// one SMILES -> row; returns 0 or a TVAE_TOK_* code.  Precedence as in the reference: the tokenizer's assertion
// sees the whole string before any id is looked up (UNMATCHED), then char_dict[...] fails on the first unknown
// token (UNKNOWN), and only a fully known row can fail on its length (TOO_LONG, data.py:31).
inline int encode_row(const Vocab& v, const char* s, int64_t n, int start_id, int end_id, int pad_id, int max_len,
                      uint8_t* row) {
  row[0] = (uint8_t)start_id;
  int k = 0;
  bool unknown = false;
  int64_t i = 0;
  while (i < n) {
    const int len = token_length(s + i, n - i);
    if (len == 0) return TVAE_TOK_UNMATCHED;
    const int id = lookup(v, s + i, len);
    if (id < 0) unknown = true;
    else if (k < max_len) row[1 + k] = (uint8_t)id;
    ++k;
    i += len;
  }
  if (unknown) return TVAE_TOK_UNKNOWN;
  if (k > max_len) return TVAE_TOK_TOO_LONG;
  if (k < max_len) {                         // '<end>' once, then '_' (nothing at all when the row is full)
    row[1 + k++] = (uint8_t)end_id;
    memset(row + 1 + k, pad_id, (size_t)(max_len - k));
  }
  return 0;
}

// ---------------------------------------------------------------------------- batch gather (device)
// One thread per output element: out[b, t] = float(ids[index[b], t]) for t < L, out[b, L] = props[index[b]].
// 1 byte read + 4 bytes written per element; rows are L <= 255 bytes, so a row's reads coalesce into 1-2 lines.
__global__ __launch_bounds__(256) void gather_batch_kernel(const uint8_t* __restrict__ ids, int64_t ld_ids,
                                                           const float* __restrict__ props,
                                                           const int64_t* __restrict__ index, int64_t first,
                                                           int64_t n_rows, int B, int L, float* __restrict__ out,
                                                           int ldo) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W = L + 1;
  if (e >= (int64_t)B * W) return;
  const int b = (int)(e / W), t = (int)(e - (int64_t)b * W);
  int64_t r = index ? index[b] : first + b;
  if (r < 0 || r >= n_rows) {                // a bad index must not become a wild read: the row reads as zeros
    out[(size_t)b * ldo + t] = 0.f;
    return;
  }
  out[(size_t)b * ldo + t] = t < L ? (float)ids[r * ld_ids + t] : (props ? props[r] : 0.f);
}

}  // namespace

extern "C" int tvae_encode_smiles(const char* text, const int64_t* offsets, int64_t n, const char* vocab,
                                  const int32_t* vocab_offsets, const int32_t* vocab_ids, int n_vocab,
                                  int start_id, int end_id, int pad_id, int max_len, uint8_t* out, int64_t ld,
                                  int n_threads, int64_t* bad_row, int* bad_code) {
  TVAE_CHECK(offsets && vocab && vocab_offsets && vocab_ids && out && (text || n == 0), TVAE_E_NULL,
             "tvae_encode_smiles: null pointer");
  TVAE_CHECK(n >= 0 && n_vocab > 0 && max_len > 0 && ld >= (int64_t)max_len + 1, TVAE_E_SHAPE,
             "tvae_encode_smiles: bad sizes n=%ld n_vocab=%d max_len=%d ld=%ld", (long)n, n_vocab, max_len, (long)ld);
  TVAE_CHECK((unsigned)start_id < 256u && (unsigned)end_id < 256u && (unsigned)pad_id < 256u, TVAE_E_SHAPE,
             "tvae_encode_smiles: start/end/pad ids must fit one byte");
  Vocab v;
  for (int i = 0; i < 256; ++i) v.single[i] = -1;
  for (int t = 0; t < n_vocab; ++t) {
    const int len = vocab_offsets[t + 1] - vocab_offsets[t];
    TVAE_CHECK(len > 0 && (unsigned)vocab_ids[t] < 256u, TVAE_E_SHAPE,
               "tvae_encode_smiles: vocabulary entry %d: empty token or id %d outside one byte", t, vocab_ids[t]);
    const char* s = vocab + vocab_offsets[t];
    if (len == 1) v.single[(unsigned char)s[0]] = (int16_t)vocab_ids[t];
    else v.multi.push_back({s, len, vocab_ids[t]});
  }
  for (int64_t i = 0; i < n; ++i)
    TVAE_CHECK(offsets[i + 1] >= offsets[i], TVAE_E_SHAPE, "tvae_encode_smiles: offsets decrease at row %ld", (long)i);

  int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  if (T < 1) T = 1;
  if (T > (n_threads > 0 ? 64 : 16)) T = n_threads > 0 ? 64 : 16;
  if ((int64_t)T > (n + 4095) / 4096) T = (int)((n + 4095) / 4096);   // a thread is not worth < 4096 rows
  if (T < 1) T = 1;
  // the reference tokenizes EVERY string before it encodes the first (data.py:26), so an untokenizable row
  // anywhere is reported before an unknown token or an over-long row
  struct Bad { int64_t unmatched = -1, other = -1; int other_code = 0; };
  std::vector<Bad> bad((size_t)T);
  auto work = [&](int t) {
    const int64_t lo = n * t / T, hi = n * (t + 1) / T;
    Bad& b = bad[(size_t)t];
    for (int64_t r = lo; r < hi; ++r) {
      const int code = encode_row(v, text + offsets[r], offsets[r + 1] - offsets[r], start_id, end_id, pad_id,
                                  max_len, out + r * ld);
      if (code == TVAE_TOK_UNMATCHED) { if (b.unmatched < 0) b.unmatched = r; }
      else if (code != 0 && b.other < 0) { b.other = r; b.other_code = code; }
    }
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  int64_t row = -1;
  int code = 0;
  for (int t = 0; t < T && row < 0; ++t)     // ranges are ascending: the first hit is the lowest row
    if (bad[(size_t)t].unmatched >= 0) { row = bad[(size_t)t].unmatched; code = TVAE_TOK_UNMATCHED; }
  for (int t = 0; t < T && row < 0; ++t)
    if (bad[(size_t)t].other >= 0) { row = bad[(size_t)t].other; code = bad[(size_t)t].other_code; }
  if (bad_row) *bad_row = row;
  if (bad_code) *bad_code = code;
  if (row >= 0) {
    tvae_set_error("tvae_encode_smiles: row %ld: %s", (long)row,
                   code == TVAE_TOK_UNMATCHED ? "text that is not a SMILES token"
                   : code == TVAE_TOK_UNKNOWN ? "token not in the vocabulary" : "more tokens than max_len");
    return TVAE_E_DATA;
  }
  return TVAE_OK;
}

extern "C" int tvae_gather_batch(const uint8_t* ids, int64_t ld_ids, const float* props, const int64_t* index,
                                 int64_t first, int64_t n_rows, int B, int L, float* out, int ldo, void* stream) {
  TVAE_CHECK(ids && out, TVAE_E_NULL, "tvae_gather_batch: null pointer");
  TVAE_CHECK(B > 0 && L > 0 && ld_ids >= L && ldo >= L + 1 && n_rows > 0, TVAE_E_SHAPE,
             "tvae_gather_batch: bad shape B=%d L=%d ld_ids=%ld ldo=%d n_rows=%ld", B, L, (long)ld_ids, ldo, (long)n_rows);
  TVAE_CHECK(index || (first >= 0 && first + B <= n_rows), TVAE_E_SHAPE,
             "tvae_gather_batch: rows [%ld, %ld) outside the dataset of %ld", (long)first, (long)(first + B), (long)n_rows);
  const int64_t total = (int64_t)B * (L + 1);
  hipLaunchKernelGGL(gather_batch_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, ids, ld_ids,
                     props, index, first, n_rows, B, L, out, ldo);
  TVAE_LAUNCH_CHECK("tvae_gather_batch");
  return TVAE_OK;
}
