// Host-side tokenizer of the reference's record wire format (no GPU work).
//
// A record is one text line of space-separated fields (tf.decode_csv with field_delim=' ',
// recommendation/code/r_model.py:146-153):
//   user country recommend_colours_1 click_colour_1 recommend_colours_2 click_colour_2 [label]
// Fields 2 and 4 are comma-separated id lists (tf.string_split(sep=',') + to_number,
// r_model.py:148-150); an empty scalar field takes the default 0, an empty list field the default
// "0,0,0,0,0,0" (the record_defaults of r_model.py:152).  Every value must be a valid int32
// (decode_csv fails otherwise).  Lines that are empty after stripping the line end are skipped.
// The reference batches the lists with tf.sparse.to_dense, which needs every list of a batch to
// have the same length; here that length is the argument bag_len and any other length is an error.
#include <stdint.h>
#include <string.h>

#include "../../include/mi355rec.h"

void mi_set_error(const char* fmt, ...);

namespace {

// parses an optionally signed decimal int32 in [p, e); false on anything else
bool parse_i32(const char* p, const char* e, int32_t* out) {
  if (p == e) return false;
  bool neg = false;
  if (*p == '-' || *p == '+') {
    neg = *p == '-';
    ++p;
    if (p == e) return false;
  }
  int64_t v = 0;
  for (; p < e; ++p) {
    if (*p < '0' || *p > '9') return false;
    v = v * 10 + (*p - '0');
    if (v > (int64_t) 1 << 32) return false;
  }
  if (neg) v = -v;
  if (v < INT32_MIN || v > INT32_MAX) return false;
  *out = (int32_t) v;
  return true;
}

}  // namespace

extern "C" int mi_parse_records(const char* text, int64_t len, int n_fields, int bag_len, int64_t max_rows,
                                int32_t* scalars, int32_t* bags, int64_t* n_rows, int64_t* consumed) {
  if (!text || len < 0 || !scalars || !bags || !n_rows || (n_fields != 6 && n_fields != 7) || bag_len < 1 ||
      max_rows < 0) {
    mi_set_error("mi_parse_records: bad arguments");
    return MI_ERR_INVALID_ARG;
  }
  const int n_scalar = n_fields - 2;
  const char* p = text;
  const char* const end = text + len;
  int64_t row = 0, line_no = 0;
  while (p < end && row < max_rows) {
    const char* nl = (const char*) memchr(p, '\n', (size_t) (end - p));
    const char* le = nl ? nl : end;       // line is [p, le)
    const char* next = nl ? nl + 1 : end;
    ++line_no;
    while (le > p && (le[-1] == '\r')) --le;
    if (le == p) {  // blank line
      p = next;
      continue;
    }
    int32_t* sc = scalars + row * n_scalar;
    int32_t* bg = bags + row * 2 * bag_len;
    int field = 0, si = 0;
    const char* f = p;
    for (;;) {
      const char* fe = (const char*) memchr(f, ' ', (size_t) (le - f));
      if (!fe) fe = le;
      if (field >= n_fields) {
        ++field;  // counted below
      } else if (field == 2 || field == 4) {
        int32_t* dst = bg + (field == 4 ? bag_len : 0);
        if (f == fe) {  // record default "0,0,0,0,0,0"
          if (bag_len != 6) {
            mi_set_error("record %lld: empty list field %d takes the default of 6 zeros, not %d entries",
                         (long long) line_no, field, bag_len);
            return MI_ERR_INVALID_ARG;
          }
          for (int k = 0; k < 6; ++k) dst[k] = 0;
        } else {
          int k = 0;
          const char* t = f;
          for (;;) {
            const char* te = (const char*) memchr(t, ',', (size_t) (fe - t));
            if (!te) te = fe;
            if (te != t) {  // string_split drops empty tokens
              int32_t v;
              if (k >= bag_len || !parse_i32(t, te, &v)) {
                mi_set_error(k >= bag_len ? "record %lld: list field %d has more than %d entries"
                                          : "record %lld: field %d is not a valid int32 list (wanted %d entries)",
                             (long long) line_no, field, bag_len);
                return MI_ERR_INVALID_ARG;
              }
              dst[k++] = v;
            }
            if (te == fe) break;
            t = te + 1;
          }
          if (k != bag_len) {
            mi_set_error("record %lld: list field %d has %d entries, expected %d", (long long) line_no, field, k,
                         bag_len);
            return MI_ERR_INVALID_ARG;
          }
        }
        ++field;
      } else {
        int32_t v = 0;
        if (f != fe && !parse_i32(f, fe, &v)) {
          mi_set_error("record %lld: field %d is not a valid int32", (long long) line_no, field);
          return MI_ERR_INVALID_ARG;
        }
        sc[si++] = v;
        ++field;
      }
      if (fe == le) break;
      f = fe + 1;
    }
    if (field != n_fields) {
      mi_set_error("record %lld: expect %d fields but have %d", (long long) line_no, n_fields, field);
      return MI_ERR_INVALID_ARG;
    }
    ++row;
    p = next;
  }
  *n_rows = row;
  if (consumed) *consumed = (int64_t) (p - text);
  return MI_OK;
}
