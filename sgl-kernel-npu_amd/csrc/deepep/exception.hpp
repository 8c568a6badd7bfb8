// Host-side error type of the MI355X deep_ep runtime.
// Mirrors the role of the reference's EPException / EP_HOST_ASSERT (csrc/deepep/exception.hpp:8-51): a failed check
// surfaces in Python as RuntimeError("Failed: Assertion error <file>:<line> ...").
#pragma once
#include <exception>
#include <sstream>
#include <string>

namespace deep_ep {

class EPException : public std::exception {
public:
    EPException(const char *kind, const char *file, int line, const std::string &what)
    {
        std::ostringstream os;
        os << "Failed: " << kind << " error " << file << ":" << line << " error message or error code is '" << what << "'";
        msg_ = os.str();
    }
    const char *what() const noexcept override { return msg_.c_str(); }

private:
    std::string msg_;
};

template <typename... A>
inline std::string ep_concat(A &&...a)
{
    std::ostringstream os;
    (os << ... << std::forward<A>(a));
    return os.str();
}

}  // namespace deep_ep

#define EP_HOST_ASSERT(cond)                                                                   \
    do {                                                                                       \
        if (!(cond)) throw deep_ep::EPException("Assertion", __FILE__, __LINE__, #cond);       \
    } while (0)

#define EP_HOST_ASSERT_S(cond, ...)                                                                               \
    do {                                                                                                          \
        if (!(cond))                                                                                              \
            throw deep_ep::EPException("Assertion", __FILE__, __LINE__, deep_ep::ep_concat("(" #cond ") ", __VA_ARGS__)); \
    } while (0)

#define HIP_CHECK(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            throw deep_ep::EPException("HIP Assertion", __FILE__, __LINE__,                                       \
                                       deep_ep::ep_concat(#expr, " -> ", hipGetErrorString(e_)));                \
    } while (0)

#define MI_EP_CHECK(expr)                                                                                 \
    do {                                                                                                  \
        int rc_ = (expr);                                                                                 \
        if (rc_ != 0)                                                                                     \
            throw deep_ep::EPException("Kernel launch", __FILE__, __LINE__, deep_ep::ep_concat(#expr, " -> ", rc_)); \
    } while (0)
