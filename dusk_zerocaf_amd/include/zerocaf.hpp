// zerocaf.hpp -- C++ host-side mirror of zerocaf's operator surface for the hot path,
// over the C ABI of libzerocaf_hip.so (include/zerocaf_hip.h).  Same names, argument
// meaning and error behaviour as the reference's Rust API:
//   FieldElement   src/backend/u64/field.rs  (Add/Sub/Neg/Mul :170-275, Square :302-315,
//                  inverse :854-925 (throws where Rust panics), from_bytes/to_bytes :563-631,
//                  sqrt_ratio_i :462-503, inv_sqrt :443-460)
//   Scalar         src/backend/u64/scalar.rs (Add/Sub/Neg/Mul :139-270, Square :272-283,
//                  from_bytes :445-467 (throws on > L-1), to_bytes :477-516, two_pow_k :525-552)
//   EdwardsPoint   src/edwards.rs (identity :381-391, Neg :440-463, Add :465-501, Sub :503-545,
//                  Mul<Scalar> :547-577, Double :579-592, compress :613-629, == :360-370)
//   CompressedEdwardsY::decompress :313-326 (std::optional = Option)
//   RistrettoPoint / CompressedRistretto  src/ristretto.rs (:96-154, :166-176, :224-425)
//   mul_by_cofactor / mul_by_pow_2  src/edwards.rs:174-191
//   AffinePoint::from :1071-1092, ProjectivePoint add/double/into :809-942/:402-417, is_valid,
//   Elligator / from_uniform_bytes  src/ristretto.rs:430-507; msm / fixed-base batches (not in the reference)
// Single-element operators are batches of one (convenience / tests); use the *_batch
// functions for throughput.  All arithmetic runs on the GPU; there is no CPU path.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/zerocaf_hip.h"

namespace zerocaf {

class Backend {
public:
    static zc_ctx* ctx()
    {
        static Backend b;
        return b.ctx_;
    }
    static void check(int rc, const char* what)
    {
        if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + zc_last_error());
    }
    static int device_count() { return zc_device_count(); }
    static std::string version() { return zc_version(); }
    // launch on the caller's hipStream_t (device buffers: calls become asynchronous on it) / back to the own stream
    static void set_stream(void* hip_stream) { check(zc_ctx_set_stream(ctx(), hip_stream, 1), "zc_ctx_set_stream"); }
    static void use_own_stream() { check(zc_ctx_set_stream(ctx(), nullptr, 0), "zc_ctx_set_stream"); }
    static void synchronize() { check(zc_ctx_synchronize(ctx()), "zc_ctx_synchronize"); }
    // the HIP device behind slot `slot` of the context (what zc_ctx_create(NULL, 0, ..) picked, or the list it was given)
    static int device(int slot = 0)
    {
        const int d = zc_ctx_device(ctx(), slot);
        if (d < 0) check(d, "zc_ctx_device");
        return d;
    }
    static int slot_count() { return zc_ctx_device_count(ctx()); }
    static void set_stream_dev(int slot, void* hip_stream) { check(zc_ctx_set_stream_dev(ctx(), slot, hip_stream, 1), "zc_ctx_set_stream_dev"); }
    // pin a long-lived host buffer: host batches from it copy asynchronously
    static void host_register(void* p, size_t bytes) { check(zc_host_register(p, bytes), "zc_host_register"); }
    static void host_unregister(void* p) { check(zc_host_unregister(p), "zc_host_unregister"); }
    // one process per GPU: RCCL communicator for the MSM exchange (rank 0 makes the id, every rank joins)
    static std::array<uint8_t, 128> comm_unique_id()
    {
        std::array<uint8_t, 128> id{};
        check(zc_comm_unique_id(id.data()), "zc_comm_unique_id");
        return id;
    }
    static void comm_init(const std::array<uint8_t, 128>& id, int rank, int world) { check(zc_comm_init(ctx(), id.data(), rank, world), "zc_comm_init"); }
    static void comm_destroy() { check(zc_comm_destroy(ctx()), "zc_comm_destroy"); }
    static int comm_size() { int r = 0; check(zc_comm_size(ctx(), &r), "zc_comm_size"); return r; }   // ncclCommCount; 0 without a communicator

private:
    Backend() { check(zc_ctx_create(nullptr, 0, &ctx_), "zc_ctx_create"); }
    ~Backend() { zc_ctx_destroy(ctx_); }
    zc_ctx* ctx_ = nullptr;
};

struct Scalar;

struct FieldElement {
    std::array<uint64_t, 5> l{};
    FieldElement() = default;
    FieldElement(std::array<uint64_t, 5> v) : l(v) {}
    explicit FieldElement(uint64_t v) { l = {v & ((1ull << 52) - 1), v >> 52, 0, 0, 0}; }   // From<u64>, field.rs:124-133
    static FieldElement zero() { return FieldElement(); }
    static FieldElement one() { return FieldElement(std::array<uint64_t, 5>{1, 0, 0, 0, 0}); }
    static FieldElement minus_one() { return FieldElement(std::array<uint64_t, 5>{671914833335276ull, 3916664325105025ull, 1367801ull, 0, 17592186044416ull}); }
    static FieldElement two_pow_k(uint64_t k)                               // field.rs:640-666 (host-side constructor, as in the reference)
    {
        if (k >= 253) throw std::domain_error("Exponent can't be greater than 260");
        FieldElement r;
        r.l[k / 52] = 1ull << (k % 52);
        return r;
    }

    FieldElement operator+(const FieldElement& b) const { FieldElement r; Backend::check(zc_fe_add(Backend::ctx(), l.data(), b.l.data(), r.l.data(), 1), "zc_fe_add"); return r; }
    FieldElement operator-(const FieldElement& b) const { FieldElement r; Backend::check(zc_fe_sub(Backend::ctx(), l.data(), b.l.data(), r.l.data(), 1), "zc_fe_sub"); return r; }
    FieldElement operator*(const FieldElement& b) const { FieldElement r; Backend::check(zc_fe_mul(Backend::ctx(), l.data(), b.l.data(), r.l.data(), 1), "zc_fe_mul"); return r; }
    FieldElement operator-() const { FieldElement r; Backend::check(zc_fe_neg(Backend::ctx(), l.data(), r.l.data(), 1), "zc_fe_neg"); return r; }
    FieldElement square() const { FieldElement r; Backend::check(zc_fe_square(Backend::ctx(), l.data(), r.l.data(), 1), "zc_fe_square"); return r; }
    FieldElement inverse() const
    {
        FieldElement r;
        uint8_t ok = 0;
        Backend::check(zc_fe_invert(Backend::ctx(), l.data(), r.l.data(), &ok, 1), "zc_fe_invert");
        if (!ok) throw std::domain_error("inverse of zero");             // field.rs:864 assert!
        return r;
    }
    FieldElement operator/(const FieldElement& b) const
    {
        FieldElement r;
        uint8_t ok = 0;
        Backend::check(zc_fe_div(Backend::ctx(), l.data(), b.l.data(), r.l.data(), &ok, 1), "zc_fe_div");
        if (!ok) throw std::domain_error("Cannot divide by zero.");           // field.rs:285
        return r;
    }
    FieldElement half() const { FieldElement r; Backend::check(zc_fe_half(Backend::ctx(), l.data(), r.l.data(), 1), "zc_fe_half"); return r; }   // Half, field.rs:317-323
    FieldElement pow(const FieldElement& e) const                                       // Pow, field.rs:325-355
    {
        FieldElement r;
        Backend::check(zc_fe_pow(Backend::ctx(), l.data(), e.l.data(), r.l.data(), 1), "zc_fe_pow");
        return r;
    }
    bool legendre_symbol() const                                                        // field.rs:703-706
    {
        uint8_t c = 0;
        Backend::check(zc_fe_legendre_symbol(Backend::ctx(), l.data(), &c, 1), "zc_fe_legendre_symbol");
        return c != 0;
    }
    bool is_positive() const                                                            // field.rs:552-557
    {
        uint8_t c = 0;
        Backend::check(zc_fe_is_positive(Backend::ctx(), l.data(), &c, 1), "zc_fe_is_positive");
        return c != 0;
    }
    bool is_even() const { return (l[0] & 1) == 0; }                                    // field.rs:534-539
    std::optional<FieldElement> mod_sqrt(bool sign) const;                              // ModSqrt, field.rs:357-441
    // (Choice, FieldElement), field.rs:462-503
    std::pair<bool, FieldElement> sqrt_ratio_i(const FieldElement& v) const
    {
        FieldElement r;
        uint8_t sq = 0;
        Backend::check(zc_fe_sqrt_ratio_i(Backend::ctx(), l.data(), v.l.data(), r.l.data(), &sq, 1), "zc_fe_sqrt_ratio_i");
        return {sq != 0, r};
    }
    std::pair<bool, FieldElement> inv_sqrt() const                                      // InvSqrt, field.rs:443-460
    {
        FieldElement r;
        uint8_t sq = 0;
        Backend::check(zc_fe_inv_sqrt(Backend::ctx(), l.data(), r.l.data(), &sq, 1), "zc_fe_inv_sqrt");
        return {sq != 0, r};
    }
    static FieldElement from_bytes(const std::array<uint8_t, 32>& b)
    {
        FieldElement r;
        Backend::check(zc_fe_from_bytes(Backend::ctx(), b.data(), r.l.data(), 1), "zc_fe_from_bytes");
        return r;
    }
    std::array<uint8_t, 32> to_bytes() const
    {
        std::array<uint8_t, 32> b{};
        Backend::check(zc_fe_to_bytes(Backend::ctx(), l.data(), b.data(), 1), "zc_fe_to_bytes");
        return b;
    }
    bool operator==(const FieldElement& o) const { return to_bytes() == o.to_bytes(); }     // src/field.rs:93-106
    bool operator!=(const FieldElement& o) const { return !(*this == o); }
    uint64_t operator[](size_t i) const { return l[i]; }
};

inline std::optional<FieldElement> FieldElement::mod_sqrt(bool sign) const
{
    FieldElement r;
    uint8_t ok = 0;
    Backend::check(zc_fe_mod_sqrt(Backend::ctx(), l.data(), sign ? 1 : 0, r.l.data(), &ok, 1), "zc_fe_mod_sqrt");
    if (!ok) return std::nullopt;
    return r;
}

struct Scalar {
    std::array<uint64_t, 5> l{};
    Scalar() = default;
    Scalar(std::array<uint64_t, 5> v) : l(v) {}
    explicit Scalar(uint64_t v) { l = {v & ((1ull << 52) - 1), v >> 52, 0, 0, 0}; }
    static Scalar zero() { return Scalar(); }
    static Scalar one() { return Scalar(std::array<uint64_t, 5>{1, 0, 0, 0, 0}); }
    static Scalar minus_one() { return Scalar(std::array<uint64_t, 5>{1129677152307298ull, 1363544697812651ull, 714439ull, 0, 2199023255552ull}); }
    static Scalar two_pow_k(uint64_t k)                                     // scalar.rs:525-552
    {
        if (k >= 250) throw std::domain_error("Exponent can't be greater than the sub-group order");
        Scalar s;
        s.l[k / 52] = 1ull << (k % 52);
        return s;
    }
    Scalar operator+(const Scalar& b) const { Scalar r; Backend::check(zc_sc_add(Backend::ctx(), l.data(), b.l.data(), r.l.data(), 1), "zc_sc_add"); return r; }
    Scalar operator-(const Scalar& b) const { Scalar r; Backend::check(zc_sc_sub(Backend::ctx(), l.data(), b.l.data(), r.l.data(), 1), "zc_sc_sub"); return r; }
    Scalar operator*(const Scalar& b) const { Scalar r; Backend::check(zc_sc_mul(Backend::ctx(), l.data(), b.l.data(), r.l.data(), 1), "zc_sc_mul"); return r; }
    Scalar operator-() const { Scalar r; Backend::check(zc_sc_neg(Backend::ctx(), l.data(), r.l.data(), 1), "zc_sc_neg"); return r; }
    Scalar square() const { Scalar r; Backend::check(zc_sc_square(Backend::ctx(), l.data(), r.l.data(), 1), "zc_sc_square"); return r; }
    Scalar half() const { Scalar r; Backend::check(zc_sc_half(Backend::ctx(), l.data(), r.l.data(), 1), "zc_sc_half"); return r; }   // Half, scalar.rs:285-291
    Scalar pow(const Scalar& e) const { Scalar r; Backend::check(zc_sc_pow(Backend::ctx(), l.data(), e.l.data(), r.l.data(), 1), "zc_sc_pow"); return r; }   // Pow, :300-322
    Scalar operator>>(uint8_t k) const { Scalar r; Backend::check(zc_sc_shr(Backend::ctx(), l.data(), k, r.l.data(), 1), "zc_sc_shr"); return r; }            // Shr<u8>, :165-182
    std::array<uint8_t, 256> into_bits() const                               // scalar.rs:352-366
    {
        std::array<uint8_t, 256> b{};
        Backend::check(zc_sc_into_bits(Backend::ctx(), l.data(), b.data(), 1), "zc_sc_into_bits");
        return b;
    }
    std::array<int8_t, 256> compute_NAF() const                              // scalar.rs:370-389
    {
        std::array<int8_t, 256> d{};
        Backend::check(zc_sc_compute_naf(Backend::ctx(), l.data(), 0, d.data(), 1), "zc_sc_compute_naf");
        return d;
    }
    std::array<int8_t, 256> compute_window_NAF(uint8_t width) const          // scalar.rs:396-415
    {
        std::array<int8_t, 256> d{};
        Backend::check(zc_sc_compute_naf(Backend::ctx(), l.data(), width, d.data(), 1), "zc_sc_compute_naf");
        return d;
    }
    bool is_even() const { return (l[0] & 1) == 0; }                         // scalar.rs:346-348
    Scalar half_without_mod() const                                         // scalar.rs:562-574
    {
        Scalar r = *this;
        uint64_t carry = 0;
        for (int i = 4; i >= 0; i--) {
            r.l[i] |= carry;
            carry = (r.l[i] & 1) << 52;
            r.l[i] >>= 1;
        }
        return r;
    }
    static Scalar from_bytes(const std::array<uint8_t, 32>& b)
    {
        Scalar r;
        uint8_t ok = 0;
        Backend::check(zc_sc_from_bytes(Backend::ctx(), b.data(), r.l.data(), &ok, 1), "zc_sc_from_bytes");
        if (!ok) throw std::domain_error("scalar bytes > L - 1");            // scalar.rs:465 assert!
        return r;
    }
    std::array<uint8_t, 32> to_bytes() const
    {
        std::array<uint8_t, 32> b{};
        Backend::check(zc_sc_to_bytes(Backend::ctx(), l.data(), b.data(), 1), "zc_sc_to_bytes");
        return b;
    }
    bool operator==(const Scalar& o) const { return to_bytes() == o.to_bytes(); }           // src/scalar.rs:78-91
    uint64_t operator[](size_t i) const { return l[i]; }
};

struct CompressedEdwardsY;
struct CompressedRistretto;

struct EdwardsPoint {
    FieldElement X, Y, Z, T;
    static EdwardsPoint identity() { return {FieldElement::zero(), FieldElement::one(), FieldElement::one(), FieldElement::zero()}; }
    void flat(uint64_t* o) const
    {
        std::memcpy(o, X.l.data(), 40);
        std::memcpy(o + 5, Y.l.data(), 40);
        std::memcpy(o + 10, Z.l.data(), 40);
        std::memcpy(o + 15, T.l.data(), 40);
    }
    static EdwardsPoint unflat(const uint64_t* p)
    {
        EdwardsPoint r;
        std::memcpy(r.X.l.data(), p, 40);
        std::memcpy(r.Y.l.data(), p + 5, 40);
        std::memcpy(r.Z.l.data(), p + 10, 40);
        std::memcpy(r.T.l.data(), p + 15, 40);
        return r;
    }
    EdwardsPoint operator+(const EdwardsPoint& q) const
    {
        uint64_t a[20], b[20], o[20];
        flat(a); q.flat(b);
        Backend::check(zc_ed_add(Backend::ctx(), a, b, o, 1), "zc_ed_add");
        return unflat(o);
    }
    EdwardsPoint operator-(const EdwardsPoint& q) const
    {
        uint64_t a[20], b[20], o[20];
        flat(a); q.flat(b);
        Backend::check(zc_ed_sub(Backend::ctx(), a, b, o, 1), "zc_ed_sub");
        return unflat(o);
    }
    EdwardsPoint operator-() const
    {
        uint64_t a[20], o[20];
        flat(a);
        Backend::check(zc_ed_neg(Backend::ctx(), a, o, 1), "zc_ed_neg");
        return unflat(o);
    }
    EdwardsPoint double_() const
    {
        uint64_t a[20], o[20];
        flat(a);
        Backend::check(zc_ed_double(Backend::ctx(), a, o, 1), "zc_ed_double");
        return unflat(o);
    }
    EdwardsPoint operator*(const Scalar& k) const                           // double_and_add, edwards.rs:102-120
    {
        uint64_t a[20], o[20];
        flat(a);
        Backend::check(zc_ed_scalar_mul(Backend::ctx(), a, k.l.data(), o, 1, ZC_SCALAR_MUL_STRICT), "zc_ed_scalar_mul");
        return unflat(o);
    }
    bool operator==(const EdwardsPoint& q) const                            // affine equality, edwards.rs:360-370
    {
        uint64_t a[20], b[20];
        uint8_t e = 0;
        flat(a); q.flat(b);
        Backend::check(zc_ed_eq(Backend::ctx(), a, b, &e, 1), "zc_ed_eq");
        return e != 0;
    }
    bool is_valid() const                                                   // ValidityCheck, edwards.rs:393-400
    {
        uint64_t a[20];
        uint8_t v = 0;
        flat(a);
        Backend::check(zc_ed_is_valid(Backend::ctx(), a, &v, 1), "zc_ed_is_valid");
        return v != 0;
    }
    inline CompressedEdwardsY compress() const;
};

// AffinePoint::from(EdwardsPoint), edwards.rs:1071-1092 (throws where the inverse of Z = 0 panics)
struct AffinePoint {
    FieldElement X, Y;
    static AffinePoint from(const EdwardsPoint& p)
    {
        uint64_t a[20], xy[10];
        uint8_t ok = 0;
        p.flat(a);
        Backend::check(zc_ed_to_affine(Backend::ctx(), a, xy, &ok, 1), "zc_ed_to_affine");
        if (!ok) throw std::domain_error("inverse of zero");
        AffinePoint r;
        std::memcpy(r.X.l.data(), xy, 40);
        std::memcpy(r.Y.l.data(), xy + 5, 40);
        return r;
    }
};

// ProjectivePoint (edwards.rs:666-998): add :809-865, double :905-942, into EdwardsPoint :402-417
struct ProjectivePoint {
    FieldElement X, Y, Z;
    void flat(uint64_t* o) const
    {
        std::memcpy(o, X.l.data(), 40);
        std::memcpy(o + 5, Y.l.data(), 40);
        std::memcpy(o + 10, Z.l.data(), 40);
    }
    static ProjectivePoint unflat(const uint64_t* p)
    {
        ProjectivePoint r;
        std::memcpy(r.X.l.data(), p, 40);
        std::memcpy(r.Y.l.data(), p + 5, 40);
        std::memcpy(r.Z.l.data(), p + 10, 40);
        return r;
    }
    ProjectivePoint operator+(const ProjectivePoint& q) const
    {
        uint64_t a[15], b[15], o[15];
        flat(a); q.flat(b);
        Backend::check(zc_proj_add(Backend::ctx(), a, b, o, 1), "zc_proj_add");
        return unflat(o);
    }
    ProjectivePoint double_() const
    {
        uint64_t a[15], o[15];
        flat(a);
        Backend::check(zc_proj_double(Backend::ctx(), a, o, 1), "zc_proj_double");
        return unflat(o);
    }
    EdwardsPoint to_extended() const
    {
        uint64_t a[15], o[20];
        flat(a);
        Backend::check(zc_proj_to_extended(Backend::ctx(), a, o, 1), "zc_proj_to_extended");
        return EdwardsPoint::unflat(o);
    }
    static ProjectivePoint identity() { ProjectivePoint r; r.Y.l[0] = 1; r.Z.l[0] = 1; return r; }   // edwards.rs:722-731
    ProjectivePoint operator-() const                                       // Neg, edwards.rs:787-807
    {
        uint64_t a[15], o[15];
        flat(a);
        Backend::check(zc_proj_neg(Backend::ctx(), a, o, 1), "zc_proj_neg");
        return unflat(o);
    }
    ProjectivePoint operator-(const ProjectivePoint& q) const               // Sub, edwards.rs:851-879
    {
        uint64_t a[15], b[15], o[15];
        flat(a); q.flat(b);
        Backend::check(zc_proj_sub(Backend::ctx(), a, b, o, 1), "zc_proj_sub");
        return unflat(o);
    }
    ProjectivePoint operator*(const Scalar& k) const                        // Mul<Scalar>, edwards.rs:881-912
    {
        uint64_t a[15], o[15];
        flat(a);
        Backend::check(zc_proj_scalar_mul(Backend::ctx(), a, k.l.data(), o, 1), "zc_proj_scalar_mul");
        return unflat(o);
    }
    bool operator==(const ProjectivePoint& q) const                         // edwards.rs:701-711
    {
        uint64_t a[15], b[15];
        uint8_t e = 0;
        flat(a); q.flat(b);
        Backend::check(zc_proj_eq(Backend::ctx(), a, b, &e, 1), "zc_proj_eq");
        return e != 0;
    }
    bool is_valid() const                                                   // edwards.rs:733-748
    {
        uint64_t a[15];
        uint8_t v = 0;
        flat(a);
        Backend::check(zc_proj_is_valid(Backend::ctx(), a, &v, 1), "zc_proj_is_valid");
        return v != 0;
    }
};
inline std::array<EdwardsPoint, 4> coset4(const EdwardsPoint& p)             // EdwardsPoint::coset4, edwards.rs:603-610
{
    uint64_t a[20], o[80];
    p.flat(a);
    Backend::check(zc_ed_coset4(Backend::ctx(), a, o, 1), "zc_ed_coset4");
    return {EdwardsPoint::unflat(o), EdwardsPoint::unflat(o + 20), EdwardsPoint::unflat(o + 40), EdwardsPoint::unflat(o + 60)};
}

struct CompressedEdwardsY {
    std::array<uint8_t, 32> bytes{};
    static CompressedEdwardsY identity() { CompressedEdwardsY c; c.bytes[0] = 1; return c; }   // edwards.rs:273-283
    std::optional<EdwardsPoint> decompress() const                          // edwards.rs:313-326
    {
        uint64_t o[20];
        uint8_t ok = 0;
        Backend::check(zc_ed_decompress(Backend::ctx(), bytes.data(), o, &ok, 1), "zc_ed_decompress");
        if (!ok) return std::nullopt;
        return EdwardsPoint::unflat(o);
    }
    bool operator==(const CompressedEdwardsY& o) const { return bytes == o.bytes; }
};
inline CompressedEdwardsY EdwardsPoint::compress() const                    // edwards.rs:613-629
{
    uint64_t a[20];
    CompressedEdwardsY c;
    uint8_t ok = 0;
    flat(a);
    Backend::check(zc_ed_compress(Backend::ctx(), a, c.bytes.data(), &ok, 1), "zc_ed_compress");
    if (!ok) throw std::domain_error("compress: point has no affine form / x^2 is not a square");   // unwrap()/assert! panics
    return c;
}

inline EdwardsPoint mul_by_pow_2(const EdwardsPoint& p, uint64_t k)         // edwards.rs:186-191
{
    uint64_t a[20], o[20];
    p.flat(a);
    Backend::check(zc_ed_mul_by_pow_2(Backend::ctx(), a, k, o, 1), "zc_ed_mul_by_pow_2");
    return EdwardsPoint::unflat(o);
}
inline EdwardsPoint mul_by_cofactor(const EdwardsPoint& p)                  // edwards.rs:174-179
{
    uint64_t a[20], o[20];
    p.flat(a);
    Backend::check(zc_ed_mul_by_cofactor(Backend::ctx(), a, o, 1), "zc_ed_mul_by_cofactor");
    return EdwardsPoint::unflat(o);
}

struct RistrettoPoint {
    EdwardsPoint p;                                                         // RistrettoPoint(pub EdwardsPoint)
    static RistrettoPoint identity() { return {EdwardsPoint::identity()}; }
    RistrettoPoint operator+(const RistrettoPoint& q) const { return {p + q.p}; }
    RistrettoPoint operator-(const RistrettoPoint& q) const { return {p + (-q.p)}; }        // ristretto.rs:291-293
    RistrettoPoint operator-() const { return {-p}; }
    RistrettoPoint double_() const { return {p.double_()}; }
    RistrettoPoint operator*(const Scalar& k) const { return {p * k}; }
    bool operator==(const RistrettoPoint& q) const                          // ristretto.rs:166-176
    {
        uint64_t a[20], b[20];
        uint8_t e = 0;
        p.flat(a); q.p.flat(b);
        Backend::check(zc_ris_eq(Backend::ctx(), a, b, &e, 1), "zc_ris_eq");
        return e != 0;
    }
    bool is_valid() const                                                   // ristretto.rs:205-222
    {
        uint64_t a[20];
        uint8_t v = 0;
        p.flat(a);
        Backend::check(zc_ris_is_valid(Backend::ctx(), a, &v, 1), "zc_ris_is_valid");
        return v != 0;
    }
    static RistrettoPoint elligator_ristretto_flavor(const FieldElement& r0)   // ristretto.rs:430-471
    {
        uint64_t o[20];
        Backend::check(zc_ris_elligator(Backend::ctx(), r0.l.data(), o, 1), "zc_ris_elligator");
        return {EdwardsPoint::unflat(o)};
    }
    static RistrettoPoint from_uniform_bytes(const std::array<uint8_t, 64>& b)   // ristretto.rs:493-507
    {
        uint64_t o[20];
        Backend::check(zc_ris_from_uniform_bytes(Backend::ctx(), b.data(), o, 1), "zc_ris_from_uniform_bytes");
        return {EdwardsPoint::unflat(o)};
    }
    inline CompressedRistretto compress() const;
};
struct CompressedRistretto {
    std::array<uint8_t, 32> bytes{};
    static CompressedRistretto identity() { return {}; }
    std::optional<RistrettoPoint> decompress() const                        // ristretto.rs:96-154
    {
        uint64_t o[20];
        uint8_t ok = 0;
        Backend::check(zc_ris_decompress(Backend::ctx(), bytes.data(), o, &ok, 1), "zc_ris_decompress");
        if (!ok) return std::nullopt;
        return RistrettoPoint{EdwardsPoint::unflat(o)};
    }
    bool operator==(const CompressedRistretto& o) const { return bytes == o.bytes; }
};
inline CompressedRistretto RistrettoPoint::compress() const                 // ristretto.rs:398-425
{
    uint64_t a[20];
    CompressedRistretto c;
    p.flat(a);
    Backend::check(zc_ris_compress(Backend::ctx(), a, c.bytes.data(), 1), "zc_ris_compress");
    return c;
}
inline RistrettoPoint operator*(const Scalar& k, const RistrettoPoint& p) { return p * k; }   // ristretto.rs:346-360

// ---- batch API (what a caller with many elements should use) -------------------------------
inline std::vector<EdwardsPoint> mul_batch(const std::vector<EdwardsPoint>& ps, const std::vector<Scalar>& ks)
{
    if (ps.size() != ks.size()) throw std::invalid_argument("mul_batch: size mismatch");
    std::vector<uint64_t> p(ps.size() * 20), k(ks.size() * 5), o(ps.size() * 20);
    for (size_t i = 0; i < ps.size(); i++) {
        ps[i].flat(&p[20 * i]);
        std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    }
    Backend::check(zc_ed_scalar_mul(Backend::ctx(), p.data(), k.data(), o.data(), ps.size(), ZC_SCALAR_MUL_STRICT), "zc_ed_scalar_mul");
    std::vector<EdwardsPoint> out(ps.size());
    for (size_t i = 0; i < ps.size(); i++) out[i] = EdwardsPoint::unflat(&o[20 * i]);
    return out;
}
inline std::vector<FieldElement> mul_batch(const std::vector<FieldElement>& a, const std::vector<FieldElement>& b)
{
    if (a.size() != b.size()) throw std::invalid_argument("mul_batch: size mismatch");
    std::vector<uint64_t> x(a.size() * 5), y(a.size() * 5), o(a.size() * 5);
    for (size_t i = 0; i < a.size(); i++) {
        std::memcpy(&x[5 * i], a[i].l.data(), 40);
        std::memcpy(&y[5 * i], b[i].l.data(), 40);
    }
    Backend::check(zc_fe_mul(Backend::ctx(), x.data(), y.data(), o.data(), a.size()), "zc_fe_mul");
    std::vector<FieldElement> out(a.size());
    for (size_t i = 0; i < a.size(); i++) std::memcpy(out[i].l.data(), &o[5 * i], 40);
    return out;
}
inline std::vector<std::optional<CompressedRistretto>> ristretto_roundtrip_mul_batch(
    const std::vector<CompressedRistretto>& enc, const std::vector<Scalar>& ks)
{
    if (enc.size() != ks.size()) throw std::invalid_argument("size mismatch");
    std::vector<uint8_t> in(enc.size() * 32), out(enc.size() * 32), ok(enc.size());
    std::vector<uint64_t> k(ks.size() * 5);
    for (size_t i = 0; i < enc.size(); i++) {
        std::memcpy(&in[32 * i], enc[i].bytes.data(), 32);
        std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    }
    Backend::check(zc_ris_roundtrip_mul(Backend::ctx(), in.data(), k.data(), out.data(), ok.data(), enc.size()), "zc_ris_roundtrip_mul");
    std::vector<std::optional<CompressedRistretto>> r(enc.size());
    for (size_t i = 0; i < enc.size(); i++)
        if (ok[i]) {
            CompressedRistretto c;
            std::memcpy(c.bytes.data(), &out[32 * i], 32);
            r[i] = c;
        }
    return r;
}

// sum_i k_i * P_i (bucket method on the GPU); equal to the fold of operator* and operator+ under ==
inline EdwardsPoint msm(const std::vector<EdwardsPoint>& ps, const std::vector<Scalar>& ks)
{
    if (ps.size() != ks.size()) throw std::invalid_argument("msm: size mismatch");
    std::vector<uint64_t> p(ps.size() * 20), k(ks.size() * 5);
    for (size_t i = 0; i < ps.size(); i++) {
        ps[i].flat(&p[20 * i]);
        std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    }
    uint64_t o[20];
    Backend::check(zc_msm(Backend::ctx(), p.data(), k.data(), ps.size(), o), "zc_msm");
    return EdwardsPoint::unflat(o);
}
// the exchange step of a sharded MSM: ((p_0 + p_1) + p_2) + ... in index order, one kernel launch
inline EdwardsPoint fold_ordered(const std::vector<EdwardsPoint>& ps)
{
    std::vector<uint64_t> p(ps.size() * 20);
    for (size_t i = 0; i < ps.size(); i++) ps[i].flat(&p[20 * i]);
    uint64_t o[20];
    Backend::check(zc_ed_fold_ordered(Backend::ctx(), p.data(), ps.size(), o), "zc_ed_fold_ordered");
    return EdwardsPoint::unflat(o);
}
// this rank's shard of a global MSM (after Backend::comm_init): all-gather of partial sums + ordered fold
inline EdwardsPoint msm_sharded(const std::vector<EdwardsPoint>& ps, const std::vector<Scalar>& ks)
{
    std::vector<uint64_t> p(ps.size() * 20), k(ks.size() * 5);
    for (size_t i = 0; i < ps.size(); i++) {
        ps[i].flat(&p[20 * i]);
        std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    }
    uint64_t o[20];
    Backend::check(zc_msm_sharded(Backend::ctx(), p.data(), k.data(), ps.size(), o), "zc_msm_sharded");
    return EdwardsPoint::unflat(o);
}
// this device's partial sum left in device memory (out_dev: 160 bytes of HIP memory)
inline void msm_partial(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out_dev)
{
    Backend::check(zc_msm_partial(Backend::ctx(), points, scalars, n, out_dev), "zc_msm_partial");
}
// k * BASEPOINT from the fixed-base table: equal to `BASEPOINT * k` under == (not limb-identical)
inline std::vector<EdwardsPoint> mul_base_batch(const std::vector<Scalar>& ks)
{
    std::vector<uint64_t> k(ks.size() * 5), o(ks.size() * 20);
    for (size_t i = 0; i < ks.size(); i++) std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    Backend::check(zc_ed_mul_base(Backend::ctx(), k.data(), o.data(), ks.size()), "zc_ed_mul_base");
    std::vector<EdwardsPoint> out(ks.size());
    for (size_t i = 0; i < ks.size(); i++) out[i] = EdwardsPoint::unflat(&o[20 * i]);
    return out;
}
// window_naf_mul (src/edwards.rs:155-171) with its table indexed correctly, one launch; width 2..7
inline std::vector<EdwardsPoint> window_naf_mul_batch(const std::vector<Scalar>& ks, unsigned width)
{
    std::vector<uint64_t> k(ks.size() * 5), o(ks.size() * 20);
    for (size_t i = 0; i < ks.size(); i++) std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    Backend::check(zc_ed_mul_base_wnaf(Backend::ctx(), k.data(), width, o.data(), ks.size()), "zc_ed_mul_base_wnaf");
    std::vector<EdwardsPoint> out(ks.size());
    for (size_t i = 0; i < ks.size(); i++) out[i] = EdwardsPoint::unflat(&o[20 * i]);
    return out;
}
// the bucket method's plan for a shard of n pairs: {c, W, affine, record payload bytes, run, segment, sort passes, window groups,
// record stride, windows per group [4], run length per group [4]}
inline std::array<int32_t, 17> msm_plan(size_t n, bool points_aligned16 = true)
{
    std::array<int32_t, 17> v{};
    Backend::check(zc_msm_plan(Backend::ctx(), n, points_aligned16 ? 1 : 0, v.data(), 17), "zc_msm_plan");
    return v;
}
// key generation: (RISTRETTO_BASEPOINT * k).compress(), identical bytes
inline std::vector<CompressedRistretto> ristretto_keygen_batch(const std::vector<Scalar>& ks)
{
    std::vector<uint64_t> k(ks.size() * 5);
    std::vector<uint8_t> o(ks.size() * 32);
    for (size_t i = 0; i < ks.size(); i++) std::memcpy(&k[5 * i], ks[i].l.data(), 40);
    Backend::check(zc_ris_mul_base_compress(Backend::ctx(), k.data(), o.data(), ks.size()), "zc_ris_mul_base_compress");
    std::vector<CompressedRistretto> out(ks.size());
    for (size_t i = 0; i < ks.size(); i++) std::memcpy(out[i].bytes.data(), &o[32 * i], 32);
    return out;
}

namespace constants {
// src/backend/u64/constants.rs:188-211
inline EdwardsPoint BASEPOINT()
{
    return {FieldElement(std::array<uint64_t, 5>{276718085098056ull, 1646536057461434ull, 2704687245600312ull, 2630386667454967ull, 13476148227069ull}),
            FieldElement(std::array<uint64_t, 5>{1303868825475266ull, 3250718520537114ull, 2702159777242978ull, 2702159776422297ull, 10555311626649ull}),
            FieldElement::one(),
            FieldElement(std::array<uint64_t, 5>{3634527586288175ull, 2006028620404053ull, 3424252198034825ull, 2478951925947079ull, 4567251727358ull})};
}
inline RistrettoPoint RISTRETTO_BASEPOINT() { return {BASEPOINT()}; }
inline Scalar L() { return Scalar(std::array<uint64_t, 5>{1129677152307299ull, 1363544697812651ull, 714439ull, 0, 2199023255552ull}); }   // :8-9
inline FieldElement EDWARDS_D() { return FieldElement(std::array<uint64_t, 5>{3304133203739795ull, 2446467598308289ull, 1534112949566882ull, 2032729967918914ull, 2313225441931ull}); }  // :86-92
}  // namespace constants

}  // namespace zerocaf
