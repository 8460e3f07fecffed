// plugin.hip — the reference's FfiPlugin boundary (include/plhip_plugin.h) over the C ABI of plhip.h.
// Host code only: Arrow C Data Interface in, plhip_group_by_agg_opts, Arrow C Data Interface out.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/plhip.h"
#include "../../include/plhip_plugin.h"

namespace {

struct Spec {
  int n_keys = 0;
  bool stable = false;
  std::vector<plhip_agg_t> aggs;
};

const char *agg_name(int kind) {
  static const char *n[] = {"sum", "mean", "min", "max", "count"};
  return kind >= 0 && kind <= 4 ? n[kind] : "?";
}

// "plhip:k=<n>;<idx>:<agg>,...[;stable]"
bool parse_spec(const char *name, Spec &sp) {
  if (!name || strncmp(name, "plhip:k=", 8) != 0) return false;
  const char *p = name + 8;
  char *end = nullptr;
  sp.n_keys = (int)strtol(p, &end, 10);
  if (end == p || sp.n_keys < 1 || *end != ';') return false;
  p = end + 1;
  while (*p && *p != ';') {
    plhip_agg_t a;
    a.column = (int)strtol(p, &end, 10);
    if (end == p || *end != ':') return false;
    p = end + 1;
    a.kind = -1;
    for (int k = 0; k <= 4; k++) {
      const size_t l = strlen(agg_name(k));
      if (!strncmp(p, agg_name(k), l) && (p[l] == ',' || p[l] == ';' || p[l] == 0)) { a.kind = k; p += l; break; }
    }
    if (a.kind < 0) return false;
    sp.aggs.push_back(a);
    if (*p == ',') p++;
  }
  if (*p == ';') sp.stable = !strcmp(p + 1, "stable");
  return !sp.aggs.empty();
}

int dtype_of_format(const char *f) {
  if (!f || !f[0] || f[1]) return -1;
  switch (f[0]) {
    case 'c': return PLHIP_I8;  case 'C': return PLHIP_U8;
    case 's': return PLHIP_I16; case 'S': return PLHIP_U16;
    case 'i': return PLHIP_I32; case 'I': return PLHIP_U32;
    case 'l': return PLHIP_I64; case 'L': return PLHIP_U64;
    case 'f': return PLHIP_F32; case 'g': return PLHIP_F64;
    default: return -1;
  }
}
const char *format_of_dtype(int dt) {
  static const char *f[] = {"c", "s", "i", "l", "C", "S", "I", "L", "f", "g"};
  return dt >= 0 && dt <= 9 ? f[dt] : "n";
}
int width_of(int dt) {
  static const int w[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8};
  return w[dt];
}
// output dtype rules of plhip_group_by_agg (series/implementations/mod.rs:167-174, aggregations/mod.rs)
int out_dtype(int kind, int src) {
  switch (kind) {
    case PLHIP_AGG_SUM: return width_of(src) < 4 ? PLHIP_I64 : src;
    case PLHIP_AGG_MEAN: return src == PLHIP_F32 ? PLHIP_F32 : PLHIP_F64;
    case PLHIP_AGG_COUNT: return PLHIP_U32;
    default: return src;
  }
}

// ---- schemas -------------------------------------------------------------------------------------------
struct SchemaPriv {
  std::string name, format;
  std::vector<ArrowSchema *> children;
};
void release_schema(ArrowSchema *s) {
  if (!s || !s->release) return;
  SchemaPriv *p = (SchemaPriv *)s->private_data;
  for (ArrowSchema *c : p->children) {
    if (c->release) c->release(c);
    free(c);
  }
  delete p;
  s->release = nullptr;
}
void fill_schema(ArrowSchema *s, const std::string &name, const std::string &format, size_t n_children) {
  SchemaPriv *p = new SchemaPriv();
  p->name = name; p->format = format;
  p->children.resize(n_children);
  for (auto &c : p->children) { c = (ArrowSchema *)calloc(1, sizeof(ArrowSchema)); }
  s->format = p->format.c_str();
  s->name = p->name.c_str();
  s->metadata = nullptr;
  s->flags = 2; // ARROW_FLAG_NULLABLE
  s->n_children = (int64_t)n_children;
  s->children = n_children ? p->children.data() : nullptr;
  s->dictionary = nullptr;
  s->release = release_schema;
  s->private_data = p;
}

bool struct_schema(const Spec &sp, const std::vector<const ArrowSchema *> &cols, ArrowSchema *out, std::string &err) {
  const size_t n_vals = cols.size() - sp.n_keys;
  std::vector<std::pair<std::string, int>> fields;
  for (int k = 0; k < sp.n_keys; k++) {
    const int dt = dtype_of_format(cols[k]->format);
    if (dt < 0) { err = std::string("unsupported key dtype (Arrow format '") + (cols[k]->format ? cols[k]->format : "") + "')"; return false; }
    fields.emplace_back(cols[k]->name ? cols[k]->name : "", dt);
  }
  for (const plhip_agg_t &a : sp.aggs) {
    if (a.kind == PLHIP_AGG_COUNT) { fields.emplace_back("count", PLHIP_U32); continue; }
    if (a.column < 0 || (size_t)a.column >= n_vals) { err = "aggregation column index out of range"; return false; }
    const ArrowSchema *v = cols[sp.n_keys + a.column];
    const int dt = dtype_of_format(v->format);
    if (dt < 0) { err = std::string("unsupported value dtype (Arrow format '") + (v->format ? v->format : "") + "')"; return false; }
    fields.emplace_back(std::string(v->name ? v->name : "") + "_" + agg_name(a.kind), out_dtype(a.kind, dt));
  }
  fill_schema(out, "group_by", "+s", fields.size());
  for (size_t i = 0; i < fields.size(); i++) fill_schema(out->children[i], fields[i].first, format_of_dtype(fields[i].second), 0);
  return true;
}

// ---- arrays --------------------------------------------------------------------------------------------
struct ArrayPriv {
  std::vector<const void *> buffers;
  std::vector<void *> owned;
  std::vector<ArrowArray *> children;
};
void release_array(ArrowArray *a) {
  if (!a || !a->release) return;
  ArrayPriv *p = (ArrayPriv *)a->private_data;
  for (ArrowArray *c : p->children) {
    if (c->release) c->release(c);
    free(c);
  }
  for (void *b : p->owned) free(b);
  delete p;
  a->release = nullptr;
}

struct SeriesPriv {
  ArrowSchema *schema;
  std::vector<ArrowArray *> arrays;
};
// frees the boxes and the schema; array CONTENTS belong to whoever moved them out (polars-ffi lib.rs:36-48)
void release_series(SeriesExport *e) {
  if (!e || !e->release) return;
  SeriesPriv *p = (SeriesPriv *)e->private_data;
  for (ArrowArray *a : p->arrays) free(a);
  if (p->schema) {
    if (p->schema->release) p->schema->release(p->schema);
    free(p->schema);
  }
  delete p;
  e->release = nullptr;
}

SeriesExport make_series(ArrowSchema *schema, ArrowArray *array) {
  SeriesPriv *p = new SeriesPriv();
  p->schema = schema;
  p->arrays.push_back(array);
  SeriesExport e;
  e.field = schema;
  e.arrays = p->arrays.data();
  e.len = 1;
  e.release = release_series;
  e.private_data = p;
  return e;
}

ArrowArray *new_array(int64_t length, size_t n_buffers, size_t n_children) {
  ArrowArray *a = (ArrowArray *)calloc(1, sizeof(ArrowArray));
  ArrayPriv *p = new ArrayPriv();
  p->buffers.assign(n_buffers, nullptr);
  p->children.resize(n_children);
  for (auto &c : p->children) c = nullptr;
  a->length = length;
  a->null_count = -1;
  a->offset = 0;
  a->n_buffers = (int64_t)n_buffers;
  a->n_children = (int64_t)n_children;
  a->buffers = p->buffers.data();
  a->children = n_children ? p->children.data() : nullptr;
  a->dictionary = nullptr;
  a->release = release_array;
  a->private_data = p;
  return a;
}

SeriesExport empty_result(const Spec *sp, const std::vector<const ArrowSchema *> *cols) {
  ArrowSchema *schema = (ArrowSchema *)calloc(1, sizeof(ArrowSchema));
  std::string err;
  if (!sp || !cols || !struct_schema(*sp, *cols, schema, err)) fill_schema(schema, "group_by", "+s", 0);
  ArrowArray *arr = new_array(0, 1, (size_t)schema->n_children);
  ArrayPriv *ap = (ArrayPriv *)arr->private_data;
  for (int64_t i = 0; i < schema->n_children; i++) ap->children[i] = new_array(0, 2, 0);
  return make_series(schema, arr);
}

// one input series as a single-chunk host column (zero-copy for one chunk, concatenated otherwise)
struct HostCol {
  plhip_column_t col;
  std::vector<uint8_t> values, validity;
};
bool import_column(const SeriesExport &s, HostCol &h, std::string &err) {
  const int dt = dtype_of_format(s.field ? s.field->format : nullptr);
  if (dt < 0) { err = "unsupported column dtype"; return false; }
  const int w = width_of(dt);
  h.col.dtype = dt;
  h.col.location = PLHIP_HOST;
  int64_t total = 0;
  bool any_valid = false;
  for (size_t c = 0; c < s.len; c++) {
    const ArrowArray *a = s.arrays[c];
    if (!a || a->n_buffers < 2) { err = "malformed primitive array"; return false; }
    total += a->length;
    if (a->buffers[0] && a->null_count != 0) any_valid = true;
  }
  if (s.len == 1) {
    const ArrowArray *a = s.arrays[0];
    h.col.values = a->buffers[1];
    h.col.validity = any_valid ? (const uint8_t *)a->buffers[0] : nullptr;
    h.col.offset = a->offset;
    h.col.length = a->length;
    return true;
  }
  h.values.resize((size_t)total * w + 8);
  if (any_valid) h.validity.assign(((size_t)total + 7) / 8 + 8, 0);
  int64_t pos = 0;
  for (size_t c = 0; c < s.len; c++) {
    const ArrowArray *a = s.arrays[c];
    if (a->length) memcpy(h.values.data() + (size_t)pos * w, (const uint8_t *)a->buffers[1] + (size_t)a->offset * w, (size_t)a->length * w);
    if (any_valid) {
      const uint8_t *bm = (const uint8_t *)a->buffers[0];
      for (int64_t i = 0; i < a->length; i++) {
        const int64_t b = a->offset + i;
        const bool ok = !bm || ((bm[b >> 3] >> (b & 7)) & 1);
        if (ok) h.validity[(size_t)(pos + i) >> 3] |= (uint8_t)(1u << ((pos + i) & 7));
      }
    }
    pos += a->length;
  }
  h.col.values = h.values.data();
  h.col.validity = any_valid ? h.validity.data() : nullptr;
  h.col.offset = 0;
  h.col.length = total;
  return true;
}

void release_inputs(const SeriesExport *inputs, size_t n) {
  for (size_t i = 0; i < n; i++) {
    SeriesExport *s = const_cast<SeriesExport *>(&inputs[i]);
    for (size_t c = 0; c < s->len; c++)
      if (s->arrays && s->arrays[c] && s->arrays[c]->release) s->arrays[c]->release(s->arrays[c]);
    if (s->release) s->release(s);
  }
}

} // namespace

void plhip_internal_set_error(const char *msg); // api.hip: leaves a message for plhip_last_error()
static void plhip_set_last_error(const char *msg) { plhip_internal_set_error(msg); }

extern "C" {

SeriesExport plhip_plugin_group_by(const SeriesExport *inputs, size_t n_inputs) {
  Spec sp;
  std::string err;
  std::vector<const ArrowSchema *> cols;
  auto fail_with = [&](const std::string &m, bool have_spec) {
    plhip_set_last_error(m.c_str());
    SeriesExport out = empty_result(have_spec ? &sp : nullptr, have_spec ? &cols : nullptr);
    release_inputs(inputs, n_inputs);
    return out;
  };
  if (!inputs || n_inputs < 2 || !inputs[0].field || !parse_spec(inputs[0].field->name, sp))
    return fail_with("plhip_plugin_group_by: the first input's name must be \"plhip:k=<n_keys>;<idx>:<agg>,...\"", false);
  if ((size_t)sp.n_keys + 1 > n_inputs) return fail_with("plhip_plugin_group_by: fewer inputs than key columns", false);
  for (size_t i = 1; i < n_inputs; i++) {
    if (!inputs[i].field) return fail_with("plhip_plugin_group_by: input without a field", false);
    cols.push_back(inputs[i].field);
  }
  ArrowSchema *schema = (ArrowSchema *)calloc(1, sizeof(ArrowSchema));
  if (!struct_schema(sp, cols, schema, err)) { free(schema); cols.clear(); return fail_with("plhip_plugin_group_by: " + err, false); }

  std::vector<HostCol> hc(n_inputs - 1);
  for (size_t i = 1; i < n_inputs; i++)
    if (!import_column(inputs[i], hc[i - 1], err)) { release_schema(schema); free(schema); return fail_with("plhip_plugin_group_by: " + err, true); }
  std::vector<plhip_column_t> kc, vc;
  for (int k = 0; k < sp.n_keys; k++) kc.push_back(hc[k].col);
  for (size_t v = sp.n_keys; v < hc.size(); v++) vc.push_back(hc[v].col);
  plhip_result_t *res = nullptr;
  const int32_t rc = plhip_group_by_agg_opts(kc.data(), (int32_t)kc.size(), vc.data(), (int32_t)vc.size(), sp.aggs.data(), (int32_t)sp.aggs.size(),
                                             sp.stable ? PLHIP_FLAG_MAINTAIN_ORDER : 0u, nullptr, &res);
  if (rc != PLHIP_OK) { // the message is already in plhip_last_error()
    release_schema(schema); free(schema);
    SeriesExport out = empty_result(&sp, &cols);
    release_inputs(inputs, n_inputs);
    return out;
  }
  const int64_t G = plhip_result_num_rows(res);
  const int32_t nc = plhip_result_num_columns(res);
  ArrowArray *arr = new_array(G, 1, (size_t)nc);
  arr->null_count = 0;
  ArrayPriv *ap = (ArrayPriv *)arr->private_data;
  for (int32_t i = 0; i < nc; i++) { // the result frame dies with `res`: the Arrow arrays get copies of their own
    plhip_column_t c;
    plhip_result_column(res, i, &c);
    ArrowArray *ch = new_array(G, 2, 0);
    ArrayPriv *cp = (ArrayPriv *)ch->private_data;
    const size_t vb = (size_t)G * width_of(c.dtype);
    void *vals = malloc(vb ? vb : 8);
    if (vb) memcpy(vals, c.values, vb);
    cp->owned.push_back(vals);
    cp->buffers[1] = vals;
    if (c.validity) {
      const size_t bb = ((size_t)G + 7) / 8;
      void *bm = malloc(bb ? bb : 8);
      if (bb) memcpy(bm, c.validity, bb);
      cp->owned.push_back(bm);
      cp->buffers[0] = bm;
    } else {
      ch->null_count = 0;
    }
    ap->children[i] = ch;
  }
  plhip_result_free(res);
  release_inputs(inputs, n_inputs);
  return make_series(schema, arr);
}

ArrowSchema plhip_plugin_group_by_field(const ArrowSchema *fields, size_t n_fields) {
  ArrowSchema out;
  memset(&out, 0, sizeof(out));
  Spec sp;
  std::string err;
  std::vector<const ArrowSchema *> cols;
  if (fields && n_fields >= 2 && parse_spec(fields[0].name, sp) && (size_t)sp.n_keys + 1 <= n_fields) {
    for (size_t i = 1; i < n_fields; i++) cols.push_back(&fields[i]);
    if (struct_schema(sp, cols, &out, err)) return out;
    plhip_set_last_error(("plhip_plugin_group_by_field: " + err).c_str());
  } else {
    plhip_set_last_error("plhip_plugin_group_by_field: the first field's name must be \"plhip:k=<n_keys>;<idx>:<agg>,...\"");
  }
  fill_schema(&out, "group_by", "+s", 0);
  return out;
}

} // extern "C"
