/*
 * rt_scene_api.h — host-side scene-description API.
 *
 * Mirrors the class names, constructor signatures, inheritance and public data
 * members that scene functions written against the reference use
 * (reference raytracer/main.cu:203-524 are the users; the classes are declared in
 * Vector3.cuh:18, Texture.cuh:7-47, Material.cuh:44-144, Hitable.cuh:23-28,
 * Surface.cuh:8-33, Sphere.cuh:12-84, Triangle.cuh:11-19, Rectangle.cuh:8-85,
 * Box.cuh:10-23, Transform.cuh:9-55, Volumes.cuh:11-25, HitableList.cuh:7-16,
 * BvhNode.cuh:48-58, Camera.cuh:20-50).
 *
 * Nothing here intersects or shades anything: the nodes are *recorders*.  A scene
 * function `new`s them exactly as it did against the reference; rt::lower()
 * (csrc/lower.cpp) then walks the graph once and emits the flat tables of
 * include/rt_scene.h, which is what the HIP kernels consume.  Device qualifiers
 * are no-ops because scene construction runs on the host.
 */
#ifndef RT_SCENE_API_H
#define RT_SCENE_API_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cfloat>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* scene functions in the reference are declared `__device__ inline void sceneN(...)` */
#ifndef __HIPCC__
#ifndef __device__
#define __device__
#endif
#ifndef __host__
#define __host__
#endif
#ifndef __global__
#define __global__
#endif
#endif

/* ------------------------------------------------------------------ arena -- */
namespace rt {
/* Every node a scene `new`s is owned by the arena that was current at the time
 * (the reference's freeWorld, main.cu:541-551, frees through wrong casts; here
 * destruction of the arena releases everything). */
class Arena {
public:
    Arena();
    ~Arena();
    void* allocate(std::size_t n);      /* raw storage (pointer arrays) */
    void* allocate_node(std::size_t n); /* storage of an rt::Node; its destructor runs with the arena */
    static Arena* current();
private:
    Arena* prev_;
    std::vector<void*> blocks_;
    std::vector<void*> nodes_;
};
struct Node {
    static void* operator new(std::size_t n);
    static void  operator delete(void*) noexcept {} /* arena-owned */
    virtual ~Node() {}
};
} // namespace rt

/* ---------------------------------------------------------------- Vector3 -- */
class Vector3 {
public:
    float e[3];
    Vector3() : e{0.f, 0.f, 0.f} {}
    Vector3(float a, float b, float c) : e{a, b, c} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float r() const { return e[0]; }
    float g() const { return e[1]; }
    float b() const { return e[2]; }
    const Vector3& operator+() const { return *this; }
    Vector3 operator-() const { return Vector3(-e[0], -e[1], -e[2]); }
    float  operator[](int i) const { return e[i]; }
    float& operator[](int i) { return e[i]; }
    Vector3& operator+=(const Vector3& o) { for (int i = 0; i < 3; ++i) e[i] += o.e[i]; return *this; }
    Vector3& operator-=(const Vector3& o) { for (int i = 0; i < 3; ++i) e[i] -= o.e[i]; return *this; }
    Vector3& operator*=(const Vector3& o) { for (int i = 0; i < 3; ++i) e[i] *= o.e[i]; return *this; }
    Vector3& operator/=(const Vector3& o) { for (int i = 0; i < 3; ++i) e[i] /= o.e[i]; return *this; }
    Vector3& operator*=(float t) { for (int i = 0; i < 3; ++i) e[i] *= t; return *this; }
    /* Vector3.cuh:157-164 multiplies by the reciprocal */
    Vector3& operator/=(float t) { float k = 1.0f / t; for (int i = 0; i < 3; ++i) e[i] *= k; return *this; }
    float squaredLength() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    float length() const { return std::sqrt(squaredLength()); }
    void  normalize() { float k = 1.0f / std::sqrt(squaredLength()); for (int i = 0; i < 3; ++i) e[i] *= k; }
};
inline Vector3 operator+(const Vector3& a, const Vector3& b) { return Vector3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline Vector3 operator-(const Vector3& a, const Vector3& b) { return Vector3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline Vector3 operator*(const Vector3& a, const Vector3& b) { return Vector3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline Vector3 operator/(const Vector3& a, const Vector3& b) { return Vector3(a.e[0] / b.e[0], a.e[1] / b.e[1], a.e[2] / b.e[2]); }
inline Vector3 operator*(float t, const Vector3& v) { return Vector3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline Vector3 operator*(const Vector3& v, float t) { return Vector3(t * v.e[0], t * v.e[1], t * v.e[2]); }
/* Vector3.cuh:95-98 divides per component (unlike operator/=) */
inline Vector3 operator/(Vector3 v, float t) { return Vector3(v.e[0] / t, v.e[1] / t, v.e[2] / t); }
inline float dot(const Vector3& a, const Vector3& b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
inline Vector3 cross(const Vector3& a, const Vector3& b)
{
    return Vector3(a.e[1] * b.e[2] - a.e[2] * b.e[1],
                   -(a.e[0] * b.e[2] - a.e[2] * b.e[0]),
                   a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline Vector3 unitVector(Vector3 v) { return v / v.length(); }

/* ------------------------------------------------- scene-construction RNG -- */
/* The reference builds its worlds with cuRAND's XORWOW (main.cu:105,200,205);
 * that stream is not reproducible off CUDA and nothing depends on it.  Scene
 * construction here draws from Philox4x32-10 (csrc/philox.h), same (0,1] range. */
struct curandState {
    unsigned long long seed, subsequence, offset;
};
void  curand_init(unsigned long long seed, unsigned long long subsequence,
                  unsigned long long offset, curandState* state);
float curand_uniform(curandState* state);

/* ---------------------------------------------------------------- textures -- */
class Texture : public rt::Node {
public:
    enum Kind { CONSTANT, CHECKER, IMAGE, NOISE };
    virtual Kind kind() const = 0;
};
class ConstantTexture : public Texture {
public:
    Vector3 colour;
    ConstantTexture() {}
    ConstantTexture(Vector3 c) : colour(c) {}
    Kind kind() const override { return CONSTANT; }
};
class CheckerTexture : public Texture {
public:
    Texture *odd, *even;
    CheckerTexture() : odd(nullptr), even(nullptr) {}
    CheckerTexture(Texture* t0, Texture* t1) : odd(t1), even(t0) {} /* (even, odd) — Texture.cuh:30 */
    Kind kind() const override { return CHECKER; }
};
class ImageTexture : public Texture {
public:
    unsigned char* data;
    int width, height;
    ImageTexture() : data(nullptr), width(0), height(0) {}
    ImageTexture(unsigned char* pixels, int A, int B) : data(pixels), width(A), height(B) {}
    Kind kind() const override { return IMAGE; }
};
/* Perlin "marble".  The reference's Texture.cuh has no noise class (it stops at ImageTexture); this is
 * the noise_texture of "Ray Tracing: The Next Week" ch. 5 in the reference's style: a Texture that owns its
 * gradient and permutation tables and, like ConstantMedium and BvhNode, takes the scene's curandState to
 * fill them (a null state = a private stream seeded 419).
 *   value(u,v,p) = (1,1,1) * 0.5 * (1 + sin(scale * p.z + 10 * turb(p, 7))) */
class NoiseTexture : public Texture {
public:
    float scale;
    float ranvec[256][3];                 /* unit gradient vectors */
    unsigned char permX[256], permY[256], permZ[256];
    NoiseTexture(float sc = 1.f, curandState* randState = nullptr);
    Kind kind() const override { return NOISE; }
};

/* --------------------------------------------------------------- materials -- */
class Material : public rt::Node {
public:
    enum Kind { LAMBERT, METAL, GLASS, LIGHT, ISOTROPIC };
    virtual Kind kind() const = 0;
};
class Lambert : public Material {
public:
    Texture* diffuse;
    Lambert(Texture* a) : diffuse(a) {}
    Kind kind() const override { return LAMBERT; }
};
class Metal : public Material {
public:
    Texture* diffuse;
    float fuzz;
    Metal(Texture* a, float f) : diffuse(a), fuzz(f < 1 ? f : 1) {} /* Material.cuh:71 */
    Kind kind() const override { return METAL; }
};
class Glass : public Material {
public:
    float refIdx;
    Glass(float ri) : refIdx(ri) {}
    Kind kind() const override { return GLASS; }
};
class DiffuseLight : public Material {
public:
    Texture* emit;
    DiffuseLight(Texture* a) : emit(a) {}
    Kind kind() const override { return LIGHT; }
};
class Isotropic : public Material {
public:
    Texture* diffuse;
    Isotropic(Texture* a) : diffuse(a) {}
    Kind kind() const override { return ISOTROPIC; }
};
/* spellings used by the book / north_star */
typedef Lambert Lambertian;
typedef Glass   Dielectric;

/* ---------------------------------------------------------------- geometry -- */
class Hitable : public rt::Node {
public:
    enum Kind { SPHERE, MOVING_SPHERE, TRIANGLE, XY_RECT, XZ_RECT, YZ_RECT, FLIP, BOX,
                TRANSLATE, ROTATE_Y, MEDIUM, LIST, BVH };
    virtual Kind kind() const = 0;
};
class Surface : public Hitable {
public:
    Material* matPtr = nullptr;
};
class FlipNormals : public Surface {
public:
    Surface*  surfacePtr;
    Material* matPtr;
    FlipNormals(Surface* p) : surfacePtr(p), matPtr(p->matPtr) { Surface::matPtr = p->matPtr; }
    Kind kind() const override { return FLIP; }
};
class Sphere : public Surface {
public:
    Vector3   centre;
    float     radius;
    Material* matPtr;
    Sphere() : radius(0.f), matPtr(nullptr) {}
    Sphere(Vector3 cen, float r, Material* m) : centre(cen), radius(r), matPtr(m) { Surface::matPtr = m; }
    Kind kind() const override { return SPHERE; }
};
class MovingSphere : public Sphere {
public:
    Vector3   centre0, centre1;
    float     time0, time1, radius;
    Material* matPtr;
    MovingSphere(Vector3 cen0, Vector3 cen1, float t0, float t1, float r, Material* m)
        : centre0(cen0), centre1(cen1), time0(t0), time1(t1), radius(r), matPtr(m) { Surface::matPtr = m; }
    Kind kind() const override { return MOVING_SPHERE; }
};
class Triangle : public Surface {
public:
    Vector3   v0, v1, v2;
    Material* matPtr;
    Triangle(Vector3 a, Vector3 b, Vector3 c, Material* m) : v0(a), v1(b), v2(c), matPtr(m) { Surface::matPtr = m; }
    Kind kind() const override { return TRIANGLE; }
};
class XYRect : public Surface {
public:
    float x0, x1, y0, y1, k;
    Material* matPtr;
    XYRect(float _x0, float _x1, float _y0, float _y1, float _k, Material* mat)
        : x0(_x0), x1(_x1), y0(_y0), y1(_y1), k(_k), matPtr(mat) { Surface::matPtr = mat; }
    Kind kind() const override { return XY_RECT; }
};
class XZRect : public Surface {
public:
    float x0, x1, z0, z1, k;
    Material* matPtr;
    XZRect(float _x0, float _x1, float _z0, float _z1, float _k, Material* mat)
        : x0(_x0), x1(_x1), z0(_z0), z1(_z1), k(_k), matPtr(mat) { Surface::matPtr = mat; }
    Kind kind() const override { return XZ_RECT; }
};
class YZRect : public Surface {
public:
    float y0, y1, z0, z1, k;
    Material* matPtr;
    YZRect(float _y0, float _y1, float _z0, float _z1, float _k, Material* mat)
        : y0(_y0), y1(_y1), z0(_z0), z1(_z1), k(_k), matPtr(mat) { Surface::matPtr = mat; }
    Kind kind() const override { return YZ_RECT; }
};
/* north_star spelling: "Rectangle" = the XY one */
typedef XYRect Rectangle;

class Box : public Surface {
public:
    Vector3 pMin, pMax;
    Box(const Vector3& p0, const Vector3& p1, Material* m) : pMin(p0), pMax(p1) { matPtr = m; }
    Kind kind() const override { return BOX; }
};
class Translate : public Surface {
public:
    Vector3  offset;
    Hitable* hitablePtr;
    Translate(Hitable* p, const Vector3& displacement) : offset(displacement), hitablePtr(p) {}
    Kind kind() const override { return TRANSLATE; }
};
class RotateY : public Hitable {
public:
    Hitable* hitablePtr;
    float sinTheta, cosTheta;
    RotateY(Hitable* p, float angle) : hitablePtr(p)
    {
        /* Transform.cuh:59-61: the degree->radian product is formed in double, then narrowed */
        float radians = (float)((M_PI / 180.f) * angle);
        sinTheta = std::sin(radians);
        cosTheta = std::cos(radians);
    }
    Kind kind() const override { return ROTATE_Y; }
};
class ConstantMedium : public Surface {
public:
    Hitable*     boundary;
    float        negInvdensity;
    Material*    matPtr;
    curandState* randState; /* kept for signature compatibility; rendering uses per-path Philox */
    ConstantMedium(Hitable* b, float d, Texture* a, curandState* localRandState)
        : boundary(b), negInvdensity(-1 / d), matPtr(new Isotropic(a)), randState(localRandState)
    { Surface::matPtr = matPtr; }
    Kind kind() const override { return MEDIUM; }
};
class HitableList : public Hitable {
public:
    Hitable** list;
    int       listSize;
    HitableList() : list(nullptr), listSize(0) {}
    HitableList(Hitable** l, int n) : list(l), listSize(n) {}
    Kind kind() const override { return LIST; }
};
class BvhNode : public Hitable {
public:
    /* BvhNode.cuh:60-90 sorts and splits in-kernel; here the node only records its
     * members and the BVH itself is built by the host builder (csrc/lower.cpp). */
    std::vector<Hitable*> members;
    float time0, time1;
    BvhNode() : time0(0.f), time1(0.f) {}
    BvhNode(Hitable** l, int n, float t0, float t1, curandState* /*state*/)
        : members(l, l + n), time0(t0), time1(t1) {}
    Kind kind() const override { return BVH; }
};

/* ------------------------------------------------------------------ camera -- */
/* Same constructor signature and the same public data members as the reference's Camera (Camera.cuh:23, :47-49:
 * lowerLeftCorner, horizontal, vertical, origin, u, v, w, lensRadius, time0, time1), so a program that reads
 * (*dCamera)->origin or ->lensRadius compiles and sees the reference's values.  The members are filled by ONE routine,
 * rt::derive_camera (csrc/lower.cpp, compiled without FMA contraction, which cites the operation order it has to keep);
 * rt::lower() copies them into the 96-byte camera block of include/rt_scene.h.  The nine arguments are kept as given
 * beside them. */
class Camera;
namespace rt { void derive_camera(Camera& cam); }
class Camera : public rt::Node {
public:
    Camera(Vector3 lookfrom, Vector3 lookat, Vector3 vup, float vfov, float aspect, float aperture,
           float focusDist, float t0, float t1)
        : lensRadius(0.f), time0(t0), time1(t1),
          eye(lookfrom), target(lookat), upHint(vup), vfovDegrees(vfov), aspectRatio(aspect), apertureDiameter(aperture),
          focusDistance(focusDist), shutterOpen(t0), shutterClose(t1)
    { rt::derive_camera(*this); }
    Vector3 lowerLeftCorner, horizontal, vertical, origin;
    Vector3 u, v, w;
    float lensRadius, time0, time1;
    /* the constructor's arguments as given */
    Vector3 eye, target, upHint;
    float vfovDegrees, aspectRatio, apertureDiameter, focusDistance, shutterOpen, shutterClose;
};

#endif /* RT_SCENE_API_H */
