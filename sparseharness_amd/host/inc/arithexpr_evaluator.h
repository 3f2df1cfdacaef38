// arithexpr_evaluator.h -- buffer-size expressions of the kernel JSON, e.g.
// "(4*v_MHeight_2*v_MWidthC_1)".  Same entry point as the reference
// (inc/arithexpr_evaluator.h, src/arithexpr_evaluator.cpp:10-30), which binds
// v_MWidthC_1 / v_MHeight_2 / v_VLength_3 and evaluates through the 38 kLoC
// exprtk header; the expressions Lift emits only use + - * / and parentheses
// over integers, so a 40-line recursive-descent evaluator is enough.
// Unparseable strings (e.g. "?", quirk A-13) evaluate to 0.
#pragma once
#include <cctype>
#include <string>

class Evaluator {
public:
  static int evaluate(const std::string &expr, int v_MWidthC_1, int v_MHeight_2, int v_VLength_3) {
    Evaluator ev(expr.c_str(), v_MWidthC_1, v_MHeight_2, v_VLength_3);
    double v = ev.sum();
    ev.ws();
    return (ev.ok && *ev.p == '\0') ? (int)v : 0;
  }

private:
  Evaluator(const char *s, int w_, int h_, int v_) : p(s), w(w_), h(h_), v(v_), ok(true) {}
  const char *p;
  int w, h, v;
  bool ok;
  void ws() { while (*p == ' ' || *p == '\t') ++p; }
  double atom() {
    ws();
    if (*p == '(') {
      ++p;
      double r = sum();
      ws();
      if (*p == ')') ++p; else ok = false;
      return r;
    }
    if (*p == '-') { ++p; return -atom(); }
    if (std::isdigit((unsigned char)*p)) {
      double r = 0;
      while (std::isdigit((unsigned char)*p)) r = r * 10 + (*p++ - '0');
      return r;
    }
    if (std::isalpha((unsigned char)*p) || *p == '_') {
      std::string id;
      while (std::isalnum((unsigned char)*p) || *p == '_') id += *p++;
      if (id == "v_MWidthC_1") return w;
      if (id == "v_MHeight_2") return h;
      if (id == "v_VLength_3") return v;
    }
    ok = false;
    return 0;
  }
  double product() {
    double r = atom();
    for (ws(); ok && (*p == '*' || *p == '/'); ws()) {
      char op = *p++;
      double rhs = atom();
      r = (op == '*') ? r * rhs : (rhs != 0 ? r / rhs : 0);
    }
    return r;
  }
  double sum() {
    double r = product();
    for (ws(); ok && (*p == '+' || *p == '-'); ws()) {
      char op = *p++;
      double rhs = product();
      r = (op == '+') ? r + rhs : r - rhs;
    }
    return r;
  }
};
