// vector_generator.h -- input-vector generators (reference:
// inc/vector_generator.h:9-61): get(ix) -> element, generate(n) -> vector.
// X generators describe the dense "x" operand, Y generators the epilogue's "y".
#pragma once
#include <vector>

template <typename T> class VectorGenerator {
public:
  virtual ~VectorGenerator() = default;
  virtual T get(int ix) = 0;
  std::vector<T> generate(int length) {
    std::vector<T> v((std::size_t)(length > 0 ? length : 0));
    for (int i = 0; i < length; i++)
      v[(std::size_t)i] = get(i);
    return v;
  }
};

template <typename T> class XVectorGenerator : public VectorGenerator<T> {};
template <typename T> class YVectorGenerator : public VectorGenerator<T> {};

template <typename T> class ConstXVectorGenerator : public XVectorGenerator<T> {
  T value;
public:
  explicit ConstXVectorGenerator(T constv) : value(constv) {}
  T get(int) override { return value; }
};

template <typename T> class ConstYVectorGenerator : public YVectorGenerator<T> {
  T value;
public:
  explicit ConstYVectorGenerator(T constv) : value(constv) {}
  T get(int) override { return value; }
};

// x0/y0 of the iterative apps (reference: app/sssp.cpp:179-209,
// app/bfs.cpp:177-207): `at_source` at index 0, `elsewhere` everywhere else.
template <typename T> class InitialDistancesGeneratorX : public XVectorGenerator<T> {
  T at_source, elsewhere;
public:
  InitialDistancesGeneratorX(T source, T other) : at_source(source), elsewhere(other) {}
  T get(int ix) override { return ix == 0 ? at_source : elsewhere; }
};
template <typename T> class InitialDistancesGeneratorY : public YVectorGenerator<T> {
  T at_source, elsewhere;
public:
  InitialDistancesGeneratorY(T source, T other) : at_source(source), elsewhere(other) {}
  T get(int ix) override { return ix == 0 ? at_source : elsewhere; }
};
