// stages.hpp -- the reference's element-wise stage structs (gpu/src/stage.rs:115-1155) over the C ABI:
// `XStage<L, R>(planner, n)` then `.encode(buffers..., scalars...)`, same names and argument order.
// The reference encodes into a caller-owned command buffer and the caller commits / waits; here encode
// enqueues on the planner's stream and `planner.sync()` is the wait.
#pragma once
#include <type_traits>
#include "ministark.hpp"

namespace ms {

template <class L, class R = L>
struct StageBase {
    Planner& pl; size_t n;
    StageBase(Planner& p, size_t n_) : pl(p), n(n_) {
        if (n == 0 || (n & (n - 1))) throw std::invalid_argument("n must be a power of two");       // stage.rs:55-59
    }
    template <class... V> void chk(const V&... v) const {
        const size_t lens[] = {v.len()...};
        for (size_t l : lens) if (l != n) throw std::invalid_argument("buffer length differs from the stage's n");
    }
};

#define MS_BINARY_STAGE(Name, OP, INTO)                                                                          \
    template <class L, class R = L> struct Name : StageBase<L, R> {                                               \
        using StageBase<L, R>::StageBase;                                                                         \
        template <bool I = INTO, typename std::enable_if<I, int>::type = 0>                                       \
        void encode(GpuVec<L>& dst, const GpuVec<L>& lhs, const GpuVec<R>& rhs, long shift = 0) {                  \
            this->chk(dst, lhs, rhs);                                                                             \
            check(ms_binary(this->pl.ctx(), OP, L::id, R::id, this->n, dst.ptr(), lhs.ptr(), rhs.ptr(), shift));   \
        }                                                                                                         \
        template <bool I = INTO, typename std::enable_if<!I, int>::type = 0>                                      \
        void encode(GpuVec<L>& lhs, const GpuVec<R>& rhs, long shift = 0) {                                        \
            this->chk(lhs, rhs);                                                                                  \
            check(ms_binary(this->pl.ctx(), OP, L::id, R::id, this->n, lhs.ptr(), lhs.ptr(), rhs.ptr(), shift));   \
        }                                                                                                         \
    };
MS_BINARY_STAGE(MulIntoStage, MS_MUL, true)        // stage.rs:115-174
MS_BINARY_STAGE(MulAssignStage, MS_MUL, false)     // stage.rs:176-233
MS_BINARY_STAGE(AddAssignStage, MS_ADD, false)     // stage.rs:393-455
MS_BINARY_STAGE(AddIntoStage, MS_ADD, true)        // stage.rs:457-521
#undef MS_BINARY_STAGE

// constants are one element of R as Montgomery words (R::words of them)
#define MS_CONST_STAGE(Name, OP, INTO)                                                                           \
    template <class L, class R = L> struct Name : StageBase<L, R> {                                               \
        using StageBase<L, R>::StageBase;                                                                         \
        template <bool I = INTO, typename std::enable_if<I, int>::type = 0>                                       \
        void encode(GpuVec<L>& dst, const GpuVec<L>& lhs, const std::vector<uint64_t>& value) {                    \
            this->chk(dst, lhs);                                                                                  \
            if (value.size() != R::words) throw std::invalid_argument("constant has the wrong number of limbs");  \
            check(ms_binary_const(this->pl.ctx(), OP, L::id, R::id, this->n, dst.ptr(), lhs.ptr(), value.data())); \
        }                                                                                                         \
        template <bool I = INTO, typename std::enable_if<!I, int>::type = 0>                                      \
        void encode(GpuVec<L>& lhs, const std::vector<uint64_t>& value) {                                          \
            this->chk(lhs);                                                                                       \
            if (value.size() != R::words) throw std::invalid_argument("constant has the wrong number of limbs");  \
            check(ms_binary_const(this->pl.ctx(), OP, L::id, R::id, this->n, lhs.ptr(), lhs.ptr(), value.data())); \
        }                                                                                                         \
    };
MS_CONST_STAGE(AddIntoConstStage, MS_ADD, true)      // stage.rs:523-579
MS_CONST_STAGE(AddAssignConstStage, MS_ADD, false)   // stage.rs:637-692
MS_CONST_STAGE(MulIntoConstStage, MS_MUL, true)      // stage.rs:694-750
MS_CONST_STAGE(MulAssignConstStage, MS_MUL, false)   // stage.rs:752-806
#undef MS_CONST_STAGE

template <class L, class R = L> struct MulPowStage : StageBase<L, R> {        // stage.rs:334-391: lhs *= rhs[(i+shift)%n]^power
    using StageBase<L, R>::StageBase;
    void encode(GpuVec<L>& lhs, const GpuVec<R>& rhs, unsigned power, long shift = 0) {
        this->chk(lhs, rhs);
        check(ms_mul_pow(this->pl.ctx(), L::id, R::id, this->n, lhs.ptr(), lhs.ptr(), rhs.ptr(), power, shift));
    }
};
template <class Dst, class Src> struct ConvertIntoStage : StageBase<Dst, Src> {   // stage.rs:581-635
    using StageBase<Dst, Src>::StageBase;
    void encode(GpuVec<Dst>& dst, const GpuVec<Src>& src) {
        this->chk(dst, src);
        check(ms_convert(this->pl.ctx(), Dst::id, Src::id, this->n, dst.ptr(), src.ptr()));
    }
};

#define MS_UNARY_STAGE(Name, OP, INTO, HAS_E)                                                                    \
    template <class F> struct Name : StageBase<F, F> {                                                            \
        using StageBase<F, F>::StageBase;                                                                         \
        template <bool I = INTO, typename std::enable_if<I, int>::type = 0>                                       \
        void encode(GpuVec<F>& dst, const GpuVec<F>& src, unsigned exponent = 0) {                                 \
            this->chk(dst, src);                                                                                  \
            check(ms_unary(this->pl.ctx(), OP, F::id, this->n, dst.ptr(), src.ptr(), HAS_E ? exponent : 0));      \
        }                                                                                                         \
        template <bool I = INTO, typename std::enable_if<!I, int>::type = 0>                                      \
        void encode(GpuVec<F>& buf, unsigned exponent = 0) {                                                       \
            this->chk(buf);                                                                                       \
            check(ms_unary(this->pl.ctx(), OP, F::id, this->n, buf.ptr(), buf.ptr(), HAS_E ? exponent : 0));       \
        }                                                                                                         \
    };
MS_UNARY_STAGE(InverseInPlaceStage, MS_INV, false, false)   // stage.rs:808-853
MS_UNARY_STAGE(NegInPlaceStage, MS_NEG, false, false)       // stage.rs:855-900
MS_UNARY_STAGE(NegIntoStage, MS_NEG, true, false)           // stage.rs:902-947
MS_UNARY_STAGE(InverseIntoStage, MS_INV, true, false)       // stage.rs:949-997
MS_UNARY_STAGE(ExpIntoStage, MS_EXP, true, true)            // stage.rs:999-1054
MS_UNARY_STAGE(ExpInPlaceStage, MS_EXP, false, true)        // stage.rs:1056-1109
#undef MS_UNARY_STAGE

template <class F> struct FillBuffStage : StageBase<F, F> {                    // stage.rs:1111-1155
    using StageBase<F, F>::StageBase;
    void encode(GpuVec<F>& dst, const std::vector<uint64_t>& value) {
        this->chk(dst);
        if (value.size() != F::words) throw std::invalid_argument("constant has the wrong number of limbs");
        check(ms_fill(this->pl.ctx(), F::id, this->n, dst.ptr(), value.data()));
    }
};

}  // namespace ms
