/*
 * plhip_plugin.h — the reference's own dynamic plug point for this path, implemented by libplhip.so.
 *
 * Polars 0.33.2 loads expression plugins with `libloading` and calls
 *     unsafe extern "C" fn(*const SeriesExport, usize) -> SeriesExport
 * (crates/polars-plan/src/dsl/function_expr/plugin.rs:33-51, FunctionExpr::FfiPlugin
 * crates/polars-plan/src/dsl/function_expr/mod.rs:241-245; Python side Expr._register_plugin,
 * py-polars/polars/expr/expr.py:9487-9543) and, for the output dtype,
 *     unsafe extern "C" fn(*const ArrowSchema, usize) -> ArrowSchema
 * (plugin.rs:53-75).  SeriesExport is `#[repr(C)]` in crates/polars-ffi/src/lib.rs:16-25; ArrowSchema /
 * ArrowArray are the Arrow C Data Interface structs (crates/nano-arrow/src/ffi).
 *
 * Ownership, as the reference implements it:
 *   inputs  — the caller `mem::forget`s its exported inputs after the call (plugin.rs:42-48): the plugin owns
 *             them.  It releases every ArrowArray (array->release) and then the SeriesExport
 *             (series->release), exactly what `import_series_buffer` + Drop do (polars-ffi lib.rs:72-100,28-34).
 *   output  — consumed by `import_series` (lib.rs:72-89): the arrays are moved out of `*arrays[i]` bit-wise and
 *             released by the importer; then `release(&out)` runs, which must free the pointer boxes and the
 *             schema but never the array contents (lib.rs:36-48).
 * There is no error channel at this ABI version (a missing symbol is an unwrap panic, plugin.rs:38): on
 * failure the plugin returns a valid, EMPTY struct series and leaves the message in plhip_last_error().
 */
#ifndef PLHIP_PLUGIN_H
#define PLHIP_PLUGIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char *format;
  const char *name;
  const char *metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema **children;
  struct ArrowSchema *dictionary;
  void (*release)(struct ArrowSchema *);
  void *private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void **buffers;
  struct ArrowArray **children;
  struct ArrowArray *dictionary;
  void (*release)(struct ArrowArray *);
  void *private_data;
};
#endif

/* crates/polars-ffi/src/lib.rs:16-25 */
typedef struct SeriesExport {
  struct ArrowSchema *field;
  struct ArrowArray **arrays; /* one per chunk */
  size_t len;
  void (*release)(struct SeriesExport *);
  void *private_data;
} SeriesExport;

/*
 * Group-by + aggregation as an expression plugin (`is_elementwise = False`: it receives whole columns).
 *   inputs[0]            : the query.  Only its NAME is read (so that the field function below sees it too):
 *                              "plhip:k=<n_keys>;<value idx>:<agg>[,<value idx>:<agg>...][;stable]"
 *                          agg in sum|mean|min|max|count; value idx counts the value columns from 0;
 *                          e.g. "plhip:k=1;0:sum,1:mean" = sum(values[0]), mean(values[1]) by keys[0].
 *   inputs[1 .. k]       : key columns      (primitive: Int8..Int64, UInt8..UInt64, Float32, Float64)
 *   inputs[k+1 .. n-1]   : value columns    (same dtypes); chunked inputs are concatenated (the reference
 *                          rechunks first, executors/group_by.rs:66)
 * Returns ONE Struct series "group_by" with G rows: the key fields (input names, input dtypes) followed by one
 * field per aggregation named "<value name>_<agg>" ("count" for count), dtypes / null rules as
 * plhip_group_by_agg.  The caller unnests it.
 */
SeriesExport plhip_plugin_group_by(const SeriesExport *inputs, size_t n_inputs);

/* plugin_field (plugin.rs:53-75): the Struct field plhip_plugin_group_by will return for these input fields. */
struct ArrowSchema plhip_plugin_group_by_field(const struct ArrowSchema *fields, size_t n_fields);

#ifdef __cplusplus
}
#endif
#endif /* PLHIP_PLUGIN_H */
