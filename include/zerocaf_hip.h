/*
 * zerocaf_hip.h -- C ABI of libzerocaf_hip.so: the batched MI355X (gfx950) backend
 * for dusk-zerocaf's hot path (FieldElement / Scalar / EdwardsPoint / Ristretto
 * arithmetic on the Sonny/Doppio curve).
 *
 * The reference (crate `zerocaf`, pure Rust) has no FFI and no batch API; its
 * path sits behind operator traits on `Copy` structs.  Each entry point below is
 * the batched form of one reference operation, cited as file:line relative to the
 * reference checkout, and is what a thin Rust `extern "C"` shim binds (see
 * INTEGRATION.md).  Results are bit-identical to the reference on canonical
 * inputs (limbs < 2^52, value < modulus).  Scalar operands of Mul<Scalar> (zc_ed_scalar_mul,
 * zc_ris_roundtrip_mul, zc_ed_mul_base, zc_msm ...) may be any pattern of five limbs < 2^52,
 * i.e. any raw `Scalar([..])`: double_and_add's is_even / half_without_mod walk all 260
 * bits, and its loop test `n != Scalar::zero()` compares 32-byte encodings (src/scalar.rs:
 * 78-91 -> to_bytes, src/backend/u64/scalar.rs:477-516), so a pattern v >= 2^256 whose low
 * 256 bits are below 2^ctz(v >> 256) stops early ([0,0,0,0,1<<50] -> identity,
 * [1,0,0,0,1<<50] -> P); this is reproduced exactly.  Limb bits >= 2^52 are ignored.
 *
 * Data layout (all arrays contiguous, caller-owned, array-of-structs exactly as
 * the reference's in-memory limbs):
 *   FieldElement / Scalar : 5 x uint64  (radix 2^52, little-endian limbs)   40 B
 *   EdwardsPoint          : X | Y | Z | T = 20 x uint64                      160 B
 *   affine point          : x | y = 10 x uint64                              80 B
 *   compressed encodings  : 32 bytes
 * Every pointer may be HOST memory or DEVICE (HIP) memory; the library detects
 * which.  Host buffers are staged over PCIe and the call returns when results are
 * in the caller's buffer.  Device buffers are used in place: the kernels are
 * enqueued on the context's stream and the call returns immediately -- order
 * later work on the same stream or call zc_ctx_synchronize().
 *
 * Return value: 0 on success, negative zc_status on error.  Where the reference
 * panics or returns None for an individual element (inverse of 0, undecodable
 * point, scalar bytes > L-1), the optional `ok` mask gets 0 for that element (1
 * otherwise), the output element is zero/identity, and the call still succeeds.
 * All functions are thread-safe per context (one context per thread or external
 * locking); there is NO CPU fallback: without a usable GPU every call fails with
 * ZC_ERR_NO_DEVICE.
 */
#ifndef ZEROCAF_HIP_H
#define ZEROCAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zc_ctx zc_ctx;

typedef enum zc_status {
    ZC_OK = 0,
    ZC_ERR_BAD_ARG = -1,     /* null pointer, bad exponent (reference: assert!/panic) */
    ZC_ERR_NO_DEVICE = -2,   /* no HIP device visible */
    ZC_ERR_HIP = -3,         /* HIP runtime error; see zc_last_error() */
    ZC_ERR_NOMEM = -4,
    ZC_ERR_MIXED_MEM = -5    /* buffers of one call live on different devices */
} zc_status;

/* zc_ed_scalar_mul flags: which reference algorithm's formula sequence is reproduced.  All
 * three give (X:Y:Z:T) limbs identical to the named reference function.                    */
#define ZC_SCALAR_MUL_STRICT 0u      /* double_and_add = Mul<Scalar>, src/edwards.rs:102-120     */
#define ZC_SCALAR_MUL_LTR_BIN 1u     /* ltr_bin_mul, src/edwards.rs:122-134 (reads bits 248..0)  */
#define ZC_SCALAR_MUL_BINARY_NAF 2u  /* binary_naf_mul, src/edwards.rs:136-153: compute_NAF step for
                                        step, also above L - 1 where it is not the integer NAF;
                                        digits >= 256 (reference: index panic) are dropped       */
/* NOT limb-exact: fixed signed windows + dedicated doubling.  The result is the same group
 * element as Mul<Scalar> (== per src/edwards.rs:360-370, identical compress()/Ristretto bytes);
 * its (X:Y:Z:T) limbs differ by a projective factor.  ~1.5x the strict throughput.          */
#define ZC_SCALAR_MUL_FAST 16u

/* ---- context -------------------------------------------------------------- */
/* devices == NULL / ndev == 0: use the current HIP device.  With ndev > 1, calls on
 * HOST buffers shard the batch into ndev contiguous ranges (independent
 * elements, no exchange step).  The library's tuning knobs (ZC_* environment variables,
 * INTEGRATION.md section 6; none is needed in production) are read HERE, once, and kept with
 * the context: contexts with different settings can live side by side, and the environment
 * of a running process changes nothing for a context that exists.               */
int zc_ctx_create(const int *devices, int ndev, zc_ctx **out);
int zc_ctx_destroy(zc_ctx *ctx);
/* The HIP device behind device slot `slot` of the context (>= 0), or ZC_ERR_BAD_ARG; zc_ctx_device_count: its slots.
 * (What zc_ctx_create(NULL, 0, ..) picked: the calling thread's current HIP device.)                                  */
int zc_ctx_device(zc_ctx *ctx, int slot);
int zc_ctx_device_count(zc_ctx *ctx);
/* external != 0: launch on the caller's hipStream_t `hip_stream` (e.g.
 * torch.cuda.current_stream().cuda_stream; NULL there means the HIP null stream) for
 * device 0 of the context.  external == 0: go back to the context's own stream.   */
int zc_ctx_set_stream(zc_ctx *ctx, void *hip_stream, int external);
/* the same for device slot `slot` (index into the `devices` array of zc_ctx_create).  A switch
 * orders everything already enqueued on the old stream before later work on the new one.       */
int zc_ctx_set_stream_dev(zc_ctx *ctx, int slot, void *hip_stream, int external);
/* Waits for the streams of every device slot.  The one asynchronous error of the library surfaces here AND at the
 * start of every later entry point that touches the device (whoever synchronised the stream, the library or the
 * caller): a wave of the windowed core (zc_ed_scalar_mul with ZC_SCALAR_MUL_FAST, zc_ris_roundtrip_mul) that timed
 * out on its table slot -- a wedged device.  Such a wave writes POISON into the rows it owned (limbs / bytes of all
 * ones, ok = 0) and sets an error word in pinned host memory; the call that sees the word returns ZC_ERR_HIP once
 * and clears it; the context stays usable.  (Host-pointer calls synchronise themselves and report it directly.)   */
int zc_ctx_synchronize(zc_ctx *ctx);
int zc_device_count(void);
/* Pin / unpin a caller-owned host buffer (hipHostRegister): host batches from pinned memory copy
 * asynchronously and skip the runtime's bounce buffers; worth it for buffers reused across calls. */
int zc_host_register(void *ptr, size_t bytes);
int zc_host_unregister(void *ptr);
const char *zc_last_error(void);
const char *zc_version(void);

/* ---- FieldElement (mod p = 2^252 + 27742317777372353535851937790883648493) --- */
/* Add: src/backend/u64/field.rs:191-207   Sub: :217-240   Neg: :170-189         */
int zc_fe_add(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int zc_fe_sub(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int zc_fe_neg(zc_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
/* Mul: field.rs:250-275 (mul_internal :741-757 + montgomery_reduce :780-813, twice) */
int zc_fe_mul(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* Square: field.rs:302-315 (square_internal :763-777) */
int zc_fe_square(zc_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
/* inverse: field.rs:854-925 (panics on 0 -> ok[i] = 0, out = 0) */
int zc_fe_invert(zc_ctx *ctx, const uint64_t *a, uint64_t *out, uint8_t *ok, size_t n);
/* Div: field.rs:277-300 (divide by 0 asserts -> ok = 0)   Half: :317-323   Pow: :325-355 (e: canonical limbs) */
int zc_fe_div(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, uint8_t *ok, size_t n);
int zc_fe_half(zc_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int zc_fe_pow(zc_ctx *ctx, const uint64_t *a, const uint64_t *e, uint64_t *out, size_t n);
/* legendre_symbol: field.rs:703-706 (Choice: 1 = residue, also for 0)   is_positive: :552-557 */
int zc_fe_legendre_symbol(zc_ctx *ctx, const uint64_t *a, uint8_t *out, size_t n);
int zc_fe_is_positive(zc_ctx *ctx, const uint64_t *a, uint8_t *out, size_t n);
/* ModSqrt (Tonelli-Shanks value): field.rs:357-441; sign = Choice (1 selects p - x); None -> ok = 0 */
int zc_fe_mod_sqrt(zc_ctx *ctx, const uint64_t *a, int sign, uint64_t *out, uint8_t *ok, size_t n);
/* from_bytes: field.rs:563-587   to_bytes: :591-631 */
int zc_fe_from_bytes(zc_ctx *ctx, const uint8_t *in32, uint64_t *out, size_t n);
int zc_fe_to_bytes(zc_ctx *ctx, const uint64_t *in, uint8_t *out32, size_t n);
/* SqrtRatioI: field.rs:462-503 (InvSqrt :443-460 is u = 1) */
int zc_fe_sqrt_ratio_i(zc_ctx *ctx, const uint64_t *u, const uint64_t *v, uint64_t *out,
                       uint8_t *was_square, size_t n);
/* InvSqrt: field.rs:443-460 -- (was_square, 1/sqrt(a)) or (0, sqrt(i/a)); (0, 0) for a = 0 */
int zc_fe_inv_sqrt(zc_ctx *ctx, const uint64_t *a, uint64_t *out, uint8_t *was_square, size_t n);

/* ---- Scalar (mod L = 2^249 + 14490550575682688738086195780655237219) --------- */
/* Add: src/backend/u64/scalar.rs:184-200  Sub: :210-237  Neg: :139-155          */
int zc_sc_add(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int zc_sc_sub(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int zc_sc_neg(zc_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
/* Mul: scalar.rs:247-270 (Montgomery path :580-652)   Square: :272-283 */
int zc_sc_mul(zc_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int zc_sc_square(zc_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
/* from_bytes: scalar.rs:445-467 (asserts <= L-1 -> ok[i] = 0)   to_bytes: :477-516 */
int zc_sc_from_bytes(zc_ctx *ctx, const uint8_t *in32, uint64_t *out, uint8_t *ok, size_t n);
int zc_sc_to_bytes(zc_ctx *ctx, const uint64_t *in, uint8_t *out32, size_t n);
/* The Scalar operations beside the default scalar-mul path (SURVEY 8a row S-x).
 * Half: scalar.rs:285-291   Pow: :300-322 (e: canonical limbs; e = 0 gives 1)
 * Shr<u8>: :165-182 (the five limbs as one 260-bit integer shifted right, no reduction) */
int zc_sc_half(zc_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int zc_sc_pow(zc_ctx *ctx, const uint64_t *a, const uint64_t *e, uint64_t *out, size_t n);
int zc_sc_shr(zc_ctx *ctx, const uint64_t *a, unsigned shift, uint64_t *out, size_t n);
/* into_bits: scalar.rs:352-366 -- 256 bytes of 0/1 per scalar, the bits of to_bytes(), least significant first
 * (what ltr_bin_mul walks, edwards.rs:122-134) */
int zc_sc_into_bits(zc_ctx *ctx, const uint64_t *a, uint8_t *bits256, size_t n);
/* width = 0: compute_NAF, scalar.rs:370-389 (digits -1/0/1; what binary_naf_mul walks, edwards.rs:136-153)
 * width = 2..7: compute_window_NAF(width), scalar.rs:396-415 with mods_2_pow_k :433-442 (odd digits in
 * (-2^(w-1), 2^(w-1)); what window_naf_mul walks, edwards.rs:155-171).  256 int8 digits per scalar, least
 * significant first, zeros behind the last.  The reference's modular `k - Scalar::from(k_i)` is reproduced,
 * wrap-around near L included. */
int zc_sc_compute_naf(zc_ctx *ctx, const uint64_t *a, unsigned width, int8_t *naf256, size_t n);

/* ---- EdwardsPoint ------------------------------------------------------------ */
/* Add: src/edwards.rs:465-501   Sub: :503-545   Double: :579-592 (= add)   Neg: :440-463 */
int zc_ed_add(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
int zc_ed_sub(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
int zc_ed_double(zc_ctx *ctx, const uint64_t *p, uint64_t *out, size_t n);
int zc_ed_neg(zc_ctx *ctx, const uint64_t *p, uint64_t *out, size_t n);
/* Mul<Scalar> = double_and_add: edwards.rs:102-120, :547-577.  k: n scalars (5 x u64) */
int zc_ed_scalar_mul(zc_ctx *ctx, const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n,
                     unsigned flags);
/* mul_by_pow_2: edwards.rs:186-191 (kexp >= 250 -> ZC_ERR_BAD_ARG, reference asserts)
 * mul_by_cofactor: :174-179                                                        */
int zc_ed_mul_by_pow_2(zc_ctx *ctx, const uint64_t *p, uint64_t kexp, uint64_t *out, size_t n);
int zc_ed_mul_by_cofactor(zc_ctx *ctx, const uint64_t *p, uint64_t *out, size_t n);
/* AffinePoint::from: edwards.rs:1071-1092 (Z == 0 -> ok = 0)   ==: :360-370 */
int zc_ed_to_affine(zc_ctx *ctx, const uint64_t *p, uint64_t *xy_out, uint8_t *ok, size_t n);
int zc_ed_eq(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint8_t *eq_out, size_t n);
/* compress: edwards.rs:613-629   decompress: :313-326 (None -> ok = 0, out = identity) */
int zc_ed_compress(zc_ctx *ctx, const uint64_t *p, uint8_t *out32, uint8_t *ok, size_t n);
int zc_ed_decompress(zc_ctx *ctx, const uint8_t *in32, uint64_t *out, uint8_t *ok, size_t n);

/* ---- Ristretto ----------------------------------------------------------------- */
/* compress: src/ristretto.rs:398-425   decompress: :96-154   ct_eq: :166-176 */
int zc_ris_compress(zc_ctx *ctx, const uint64_t *p, uint8_t *out32, size_t n);
int zc_ris_decompress(zc_ctx *ctx, const uint8_t *in32, uint64_t *out, uint8_t *ok, size_t n);
int zc_ris_eq(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint8_t *eq_out, size_t n);
/* fused decompress -> Mul<Scalar> -> compress (ristretto.rs:96-154, :330-392, :398-425);
 * undecodable input -> ok = 0, out32 = 0.  Outputs are encodings, which depend only on the
 * group element: bit-identical to the reference whichever scalar-mul schedule runs inside.   */
int zc_ris_roundtrip_mul(zc_ctx *ctx, const uint8_t *in32, const uint64_t *k, uint8_t *out32,
                         uint8_t *ok, size_t n);

/* ---- "next" rows of the scope table ----------------------------------- */
/* ValidityCheck for EdwardsPoint: src/edwards.rs:393-400, :733-748 (curve equation)        */
int zc_ed_is_valid(zc_ctx *ctx, const uint64_t *p, uint8_t *valid_out, size_t n);
/* ValidityCheck for RistrettoPoint: src/ristretto.rs:205-222 (order exactly L, on curve)   */
int zc_ris_is_valid(zc_ctx *ctx, const uint64_t *p, uint8_t *valid_out, size_t n);
/* elligator_ristretto_flavor: src/ristretto.rs:430-471 (r0: FieldElement limbs, used as given) */
int zc_ris_elligator(zc_ctx *ctx, const uint64_t *r0, uint64_t *out, size_t n);
/* from_uniform_bytes: src/ristretto.rs:493-507 (in64: n x 64 bytes)                         */
int zc_ris_from_uniform_bytes(zc_ctx *ctx, const uint8_t *in64, uint64_t *out, size_t n);
/* ProjectivePoint (X|Y|Z = 15 x uint64): Add src/edwards.rs:809-834, Double :915-942,
 * From<ProjectivePoint> for EdwardsPoint :402-417                                           */
int zc_proj_add(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
int zc_proj_double(zc_ctx *ctx, const uint64_t *p, uint64_t *out, size_t n);
int zc_proj_to_extended(zc_ctx *ctx, const uint64_t *p, uint64_t *out, size_t n);
/* Neg src/edwards.rs:787-807   Sub :851-879 (self + (-other))   == :701-711 (affine images; Z = 0, where the
 * reference's inverse() panics, compares unequal)   is_valid :733-748
 * Mul<Scalar> :881-912 = double_and_add :102-120 over the projective Add / dedicated Double, identity (0, 1, 1) */
int zc_proj_neg(zc_ctx *ctx, const uint64_t *p, uint64_t *out, size_t n);
int zc_proj_sub(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
int zc_proj_eq(zc_ctx *ctx, const uint64_t *p, const uint64_t *q, uint8_t *eq, size_t n);
int zc_proj_is_valid(zc_ctx *ctx, const uint64_t *p, uint8_t *valid, size_t n);
int zc_proj_scalar_mul(zc_ctx *ctx, const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n);
/* EdwardsPoint::coset4: src/edwards.rs:603-610 with FOUR_COSET_GROUP (backend/u64/constants.rs:141-189):
 * out4 = n x 4 points [P, P + C0, P + C1, P + C2], each sum the unified addition :465-489 */
int zc_ed_coset4(zc_ctx *ctx, const uint64_t *p, uint64_t *out4, size_t n);

/* Fixed-base multiplication of BASEPOINT (constants.rs:188-211) with a precomputed comb table:
 * the correct counterpart of the reference's window_naf_mul (src/edwards.rs:155-171, which
 * mis-indexes its table).  zc_ed_mul_base: k*B equal to `&BASEPOINT * &k` under == (not
 * limb-identical).  zc_ris_mul_base_compress: (RISTRETTO_BASEPOINT * k).compress() -- key
 * generation; the 32-byte outputs are bit-identical to the reference's.                     */
int zc_ed_mul_base(zc_ctx *ctx, const uint64_t *k, uint64_t *out, size_t n);
int zc_ris_mul_base_compress(zc_ctx *ctx, const uint64_t *k, uint8_t *out32, size_t n);
/* window_naf_mul itself, src/edwards.rs:155-171, in ONE launch, with its two defects repaired: digit d selects entry
 * (|d| + 1) / 2 of BASEPOINT_ODD_MULTIPLES_TABLE (backend/u64/constants.rs:216-972: entry j = (2j - 1) B; the reference
 * indexes with d itself), and all 256 digits of compute_window_NAF(width) (backend/u64/scalar.rs:396-415, reproduced
 * literally) are read, not only 249..0.  out = (sum_i d_i 2^i) B: `&BASEPOINT * &k` under == for every canonical k.
 * width 2..7 (the digits are i8), else ZC_ERR_BAD_ARG.  The table is rebuilt on the device (cached affine records).   */
int zc_ed_mul_base_wnaf(zc_ctx *ctx, const uint64_t *k, unsigned width, uint64_t *out, size_t n);

/* ---- multi-scalar multiplication (not in the reference: sum_i k_i * P_i) -------- */
/* out_point: one EdwardsPoint (HOST memory), equal to the reference's
 * sum of `&P_i * &k_i` as a group element (compare with ==, i.e. affine/compressed). */
int zc_msm(zc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, size_t n,
           uint64_t *out_point);
/* The exchange step of a sharded MSM (BASELINE configs[4]), inside the library.
 * zc_msm over a context with several device slots already gathers the per-device partial sums
 * with peer copies and folds them on device slot 0.  For one process per GPU:
 *   zc_msm_partial      this device's sum, left in DEVICE memory (160 bytes), asynchronous
 *   zc_ed_fold_ordered  ((p_0 + p_1) + ...) in index order, unified add src/edwards.rs:465-489,
 *                       ONE kernel launch; host or device pointers
 *   zc_comm_*           an RCCL communicator owned by the context (librccl is opened on demand):
 *                       rank 0 calls zc_comm_unique_id, the 128 bytes reach the other ranks by any
 *                       host transport, every rank calls zc_comm_init
 *   zc_comm_size        the rank count RCCL itself reports for that communicator (ncclCommCount;
 *                       0 without one): what a scaling record quotes to show N ranks took part
 *   zc_msm_sharded      local bucket method -> ncclAllGather of the 160-byte partials over xGMI ->
 *                       ordered fold -> out_point (HOST memory), identical limbs on every rank.
 *                       (Point addition is not an ncclRedOp_t: all-gather + fold, not all-reduce.)
 *                       A rank whose local part fails still joins the collective (with a poison
 *                       record) and EVERY rank returns an error: nobody is left waiting.
 *   zc_msm_plan         a query, no device work: what the bucket method would do for a shard of n pairs on this
 *                       context.  Writes min(nout, 17) entries, nout >= 8: {window bits c (0: below the bucket
 *                       threshold, n scalar multiplications + folds), windows W, 1 = affine records and
 *                       7-multiplication additions / 0 = projective and 8, PAYLOAD bytes of a gathered record (112 /
 *                       128), run length of the bucket-sum kernel (window groups: the top group's), buckets per
 *                       reduction segment, sort passes, window groups G, record STRIDE in bytes, windows per group
 *                       [4] (top group first), run length per group [4]}.  What a roofline record counts its useful
 *                       work from (bench.py); points_aligned16: whether the point array is 16-byte aligned.
 *                       A ZC_MSM_GROUPS split that does not add up to the shard's W windows makes zc_msm* and this
 *                       query fail with ZC_ERR_BAD_ARG (never silently one group).                                     */
int zc_msm_partial(zc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, size_t n,
                   uint64_t *out_dev_point);
int zc_msm_plan(zc_ctx *ctx, size_t n, int points_aligned16, int32_t *out, int nout);
int zc_ed_fold_ordered(zc_ctx *ctx, const uint64_t *parts, size_t count, uint64_t *out);
int zc_comm_unique_id(uint8_t *id_out128);
int zc_comm_init(zc_ctx *ctx, const uint8_t *id128, int rank, int world);
int zc_comm_destroy(zc_ctx *ctx);
int zc_comm_size(zc_ctx *ctx, int *ranks);
int zc_msm_sharded(zc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, size_t n_local,
                   uint64_t *out_point);

#ifdef __cplusplus
}
#endif
#endif /* ZEROCAF_HIP_H */
