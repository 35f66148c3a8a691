/* A plain-C99 client of include/fcp_hip.h: what a cgo / JNI / dlsym binding of the drop-in boundary
 * compiles against.  Builds a two-column plan (one GatherV2 column, one SparseSegmentMean column with CSR
 * offsets) without a device (FCP_FLAG_HOST_ONLY), packs a request with fcp_concat_inputs the way
 * Addons>ConcatInputs does (concat_inputs_ops.cc:42-77), asks the plan for its layout and arena size, runs
 * the placement gate, and checks that computing without a device fails loudly.  No GPU needed.
 * gcc -std=c99 -Wall -Wextra -Werror -pedantic -I include tests/native/abi_c_client.c -L recom_amd -lfcp_hip */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fcp_hip.h"

#define CHECK(cond)                                                                          \
  do {                                                                                       \
    if (!(cond)) {                                                                           \
      fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond, fcp_last_error()); \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

int main(void) {
  fcp_column_desc_t cols[2];
  int32_t ranks[3] = {1, 1, 1}, esz[3] = {8, 8, 4};
  fcp_plan_desc_t d;
  fcp_plan_t *plan = NULL;
  int32_t n_columns = 0, n_groups = 0, n_host = 0, n_dev = 0, n_sym = 0, width = 0, off = -1;
  int64_t ids0[4] = {3, 1, 4, 1}, ids1[5] = {9, 2, 6, 5, 3};
  int32_t csr[5] = {0, 2, 2, 3, 5};
  int64_t dims0[1] = {4}, dims1[1] = {5}, dims2[1] = {5};
  fcp_host_tensor_t in[3];
  int64_t blob_bytes = 0, arena = 0, shard_bytes = 0, max_table = 0;
  int32_t rank_sum = 0, offsets[3], shapes[3], symbols[1] = {4};
  unsigned char blob[128];
  int64_t table_bytes[2];
  fcp_placement_t place;
  fcp_process_args_t args;

  CHECK(fcp_abi_version() == FCP_ABI_VERSION);
  memset(cols, 0, sizeof(cols));
  cols[0].form = FCP_FORM_GATHER;
  cols[0].dim = 8;
  cols[0].id_source = FCP_IDS_I64;
  cols[0].vocab = 10;
  cols[0].table_input = 0;
  cols[0].ids_input = 0;
  cols[0].seg_input = -1;
  cols[0].seg_kind = FCP_SEG_NONE;
  cols[0].seg_stride = 1;
  cols[0].rows_source = FCP_ROWS_FROM_IDS;
  cols[0].concat_slot = 0;
  cols[1].form = FCP_FORM_SEGMENT_REDUCE;
  cols[1].combiner = FCP_COMBINER_MEAN;
  cols[1].dim = 4;
  cols[1].id_source = FCP_IDS_I64;
  cols[1].vocab = 10;
  cols[1].table_input = 1;
  cols[1].ids_input = 1;
  cols[1].seg_input = 2;
  cols[1].seg_kind = FCP_SEG_CSR_I32;
  cols[1].seg_stride = 1;
  cols[1].rows_source = FCP_ROWS_FROM_SYMBOL;
  cols[1].rows_arg = 0;
  cols[1].concat_slot = 1;
  memset(&d, 0, sizeof(d));
  d.abi_version = FCP_ABI_VERSION;
  d.n_columns = 2;
  d.columns = cols;
  d.n_host_inputs = 3;
  d.host_input_ranks = ranks;
  d.host_input_elem_sizes = esz;
  d.n_device_inputs = 2;
  d.n_groups = 1;
  d.n_symbols = 1;
  d.layout = FCP_LAYOUT_CONCAT;
  d.shard_rank = 0;
  d.shard_world = 1;
  d.flags = FCP_FLAG_HOST_ONLY;
  CHECK(fcp_plan_create(&d, &plan) == FCP_OK && plan != NULL);
  CHECK(fcp_plan_counts(plan, &n_columns, &n_groups, &n_host, &n_dev, &n_sym) == FCP_OK);
  CHECK(n_columns == 2 && n_groups == 1 && n_host == 3 && n_dev == 2 && n_sym == 1);
  CHECK(fcp_plan_group_width(plan, 0, &width) == FCP_OK && width == 12);
  CHECK(fcp_plan_column_offset(plan, 1, &off) == FCP_OK && off == 8);

  in[0].data = ids0; in[0].elem_size = 8; in[0].rank = 1; in[0].dims = dims0;
  in[1].data = ids1; in[1].elem_size = 8; in[1].rank = 1; in[1].dims = dims1;
  in[2].data = csr;  in[2].elem_size = 4; in[2].rank = 1; in[2].dims = dims2;
  CHECK(fcp_concat_inputs_sizes(in, 3, &blob_bytes, &rank_sum) == FCP_OK && blob_bytes == 32 + 40 + 20 && rank_sum == 3);
  CHECK(fcp_concat_inputs(in, 3, blob, (int64_t)sizeof(blob), offsets, shapes) == FCP_OK);
  CHECK(offsets[0] == 0 && offsets[1] == 32 && offsets[2] == 72 && shapes[0] == 4 && shapes[1] == 5 && shapes[2] == 5);
  CHECK(memcmp(blob + 32, ids1, sizeof(ids1)) == 0 && memcmp(blob + 72, csr, sizeof(csr)) == 0);
  /* 4 rows x 12 floats, 128-byte aligned (alignmem, cuda_emitter.cc:967-969); CSR input needs no scratch */
  CHECK(fcp_plan_arena_bytes(plan, shapes, symbols, &arena) == FCP_OK && arena == 256);
  CHECK(fcp_plan_table_bytes(plan, &shard_bytes, &max_table) == FCP_OK && shard_bytes == 10 * 8 * 4 + 10 * 4 * 4 &&
        max_table == 10 * 8 * 4);

  /* the placement gate: two 200 GB tables on GPUs of 288 GB */
  table_bytes[0] = table_bytes[1] = 200000000000LL;
  CHECK(fcp_placement_decide(table_bytes, 2, 288000000000LL, 8000000000LL, 1, FCP_PLACE_COLUMN_SHARD, &place) ==
            FCP_ERR_UNSUPPORTED && place.min_world == 2);
  CHECK(fcp_placement_decide(table_bytes, 2, 288000000000LL, 8000000000LL, 8, FCP_PLACE_COLUMN_SHARD, &place) == FCP_OK &&
        place.mode == FCP_PLACE_COLUMN_SHARD && place.bytes_per_gpu == 200000000000LL);
  table_bytes[0] = 100000000000LL;
  table_bytes[1] = 20000000000LL;
  CHECK(fcp_placement_decide(table_bytes, 2, 288000000000LL, 8000000000LL, 8, FCP_PLACE_COLUMN_SHARD, &place) == FCP_OK &&
        place.mode == FCP_PLACE_REPLICATE);

  /* no device behind this plan: computing fails loudly, there is no CPU fallback */
  memset(&args, 0, sizeof(args));
  CHECK(fcp_process_feature_columns(plan, &args, NULL) == FCP_ERR_NO_DEVICE);
  { /* the serving-mode entry points (round 4) from plain C: a plan without a device refuses them, arguments are checked */
    double serial_us = 0.0, lanes_us = 0.0;
    int32_t verdict = 7;
    CHECK(fcp_plan_set_private_streams(plan, 3, FCP_PRIVATE_ALWAYS | FCP_PRIVATE_NO_VERIFY) == FCP_ERR_NO_DEVICE);
    CHECK(fcp_plan_set_private_streams(plan, 3, 1u << 9) == FCP_ERR_INVALID_ARGUMENT);
    CHECK(fcp_plan_probe_private_streams(plan, NULL, 24, 80, 1, &serial_us, &lanes_us) == FCP_ERR_NO_DEVICE);
    CHECK(fcp_plan_private_streams_verdict(plan, NULL, &verdict) == FCP_OK && verdict == -1);
    { /* round 5: verification at warm-up, the supervisor's record */
      fcp_private_streams_stats_t st;
      verdict = 7;
      CHECK(fcp_plan_verify_private_streams(plan, NULL, 50, &verdict) == FCP_ERR_NO_DEVICE && verdict == -1);
      CHECK(fcp_plan_verify_private_streams(NULL, NULL, 50, NULL) == FCP_ERR_INVALID_ARGUMENT);
      CHECK(fcp_plan_private_streams_stats(plan, &st) == FCP_OK && st.lane_requests == 0 && st.demoted == 0 && st.supervised_stream == NULL);
      CHECK(fcp_plan_private_streams_stats(plan, NULL) == FCP_ERR_INVALID_ARGUMENT);
    }
    CHECK(fcp_plan_set_request_order(plan, FCP_ORDER_INPUTS_READY) == FCP_OK);
    CHECK(fcp_plan_set_request_order(plan, 9) == FCP_ERR_INVALID_ARGUMENT);
    CHECK(fcp_result_wait(NULL, NULL) == FCP_ERR_INVALID_ARGUMENT);
    CHECK(fcp_result_synchronize(blob) == FCP_OK); /* nothing pending for an address no request ever used */
  }
  { /* round 5: the stager's record; a bad flag combination is refused before anything touches a device */
    fcp_stager_stats_t ss;
    fcp_stager_t *st = NULL;
    CHECK(fcp_stager_stats(NULL, &ss) == FCP_ERR_INVALID_ARGUMENT);
    CHECK(fcp_stager_create_ex(0, 1 << 20, 4, 4, 2, 2, FCP_STAGER_COPY_KERNEL | FCP_STAGER_COPY_SDMA, &st) == FCP_ERR_INVALID_ARGUMENT && st == NULL);
    CHECK(fcp_stager_create_ex(0, 1 << 20, 4, 4, 2, 2, 1u << 7, &st) == FCP_ERR_INVALID_ARGUMENT);
  }
  CHECK(strlen(fcp_status_string(FCP_ERR_NO_DEVICE)) > 0);
  CHECK(fcp_plan_destroy(plan) == FCP_OK);
  puts("abi_c_client ok");
  return 0;
}
